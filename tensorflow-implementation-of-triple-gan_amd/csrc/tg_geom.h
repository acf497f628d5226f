// Internal: the rules shared by the launchers (igemm.hip) and the host-side geometry entry points (geom.cpp) — one copy each.
#pragma once
#include <stdint.h>
#include "../../include/tg_kernels.h"

namespace tg {

// tile of an implicit-GEMM launch (cost model in geom.cpp); false when no candidate tile fits (COLSUM segments shorter than every tile)
bool igemm_pick_tile(const tg_igemm_desc* descs, int n_desc, bool colsum, const int32_t* seg_rows, int nseg, bool bf16, int* bm_out, int* bn_out);
// Work units of an implicit-GEMM launch (igemm.hip): tiles of T = m_tiles * n_tiles per sub-problem, sub-problem s with nk[s] K-tiles
// (kernel order: longest first).  ks[s] > 1 cuts every tile of s (n_sub > 1) or, for one sub-problem, the tiles [nfull, T) into ks K
// segments whose partial sums go to caller-owned scratch (ws_bytes) and are added up by a fix-up launch of n_fix workgroups.
struct IgemmSched {
  int n_units, n_fix, nfull, ks[4], pat_len, n_pat_split, n_split_sub;
  int8_t pat_sub[16], pat_k[16], pat_slot[16], split_sub[4], first_slot[4];
  int64_t ws_bytes;
};
// allow_split = false (no scratch): one unit per tile.  slots: workgroups resident at once (2 per compute unit).
void igemm_schedule(int n_sub, const int* nk, int64_t tiles, int bm, int bn, int slots, bool allow_split, IgemmSched* out);
// kernel order of the sub-problems of a launch: longest (most taps) first
void igemm_sub_order(const tg_igemm_desc* descs, int n_desc, int* order);
// channel tile of the filter-gradient kernel for a dimension of n (rows: ld_in, columns: c_out)
int wgrad_tile(int n);
// TG_IGEMM_EFF64 / TG_IGEMM_TILE tuning aids: read once when the library is loaded
void igemm_tuning_from_env();

}  // namespace tg
