// Internal: the rules shared by the launchers (igemm.hip) and the host-side geometry entry points (geom.cpp) — one copy each.
#pragma once
#include <stdint.h>
#include "../../include/tg_kernels.h"

namespace tg {

// tile of an implicit-GEMM launch (cost model in geom.cpp); false when no candidate tile fits (COLSUM segments shorter than every tile)
bool igemm_pick_tile(const tg_igemm_desc* descs, int n_desc, bool colsum, const int32_t* seg_rows, int nseg, bool bf16, int* bm_out, int* bn_out);
// channel tile of the filter-gradient kernel for a dimension of n (rows: ld_in, columns: c_out)
int wgrad_tile(int n);
// TG_IGEMM_EFF64 / TG_IGEMM_TILE tuning aids: read once when the library is loaded
void igemm_tuning_from_env();

}  // namespace tg
