// MFMA implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate.
//
//   tg_igemm_f32 : out[p,n] = act(sum_{t,c} in[pix(p,t),c] * w[n,t,c] + bias[n])
//   tg_wgrad_f32 : slab[s][t][c][n] = sum_{p in split s} in[pix(p,t),c] * dout[p,n]
//
// One kernel family serves conv fwd (SAME/VALID, stride 1/2), conv input-gradient, 5x5 s2 transposed conv
// (the four output parities as sub-problems of ONE launch), dense / NiN / ZCA (1 tap) — the geometry lives in tg_igemm_desc.
// Template variants: COLSUM (tg_igemm_colsum_*: the mean-only-BN column sums of the output accumulated in the epilogue, lane =
// channel operand order) and BF16 (tg_*_bf16: operands rounded to bf16 on the way LDS -> registers, v_mfma_f32_32x32x16_bf16).
//
// CDNA4 mapping
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the fp32 roofline, 157 TFLOP/s chip).
//    Lane l supplies A[row l&31][k l>>5], B[k l>>5][col l&31]; D: col = l&31, row = (r&3)+8(r>>2)+4(l>>5).
//  * the reduction index is permuted so that one ds_read_b128 feeds four MFMA k-steps: lane half h of
//    k-group g supplies k = 8g+4h+s in step s for BOTH operands (the sum over k is order-free).
//  * LDS tiles are [row][32 k] with a 16-B row pad (stride 36 floats): ds_read_b128 and ds_write_b128 are
//    bank-conflict free for the 16-lane / 8-lane groups of gfx950.
//  * register-staged double buffering: global loads of tile i+1 are issued before the 64 MFMAs of tile i and
//    written to the other LDS buffer after them; one barrier per K-tile; 2 workgroups per CU (76 KB LDS each).
//  * NHWC gathers are 16 B per lane, 128 B contiguous per 8 lanes, through buffer descriptors: an out-of-image tap is an
//    out-of-range offset that the hardware turns into zeros (no branch, no select); the K-tile position rides in the scalar
//    offset operand.  The fp32 MFMA executes on the vector ALUs, so the K loop carries next to no VALU address arithmetic.
//  * workgroup order is XCD-aware in both kernels (tiles that share operand rows meet in one XCD's L2).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "tg_common.h"
#include "tg_device.h"
#include "tg_geom.h"
#include "tg_conv3x3_bf16.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// bf16 variant (tg_*_bf16 entry points): tensors stay fp32 in HBM and in LDS; every MFMA operand fragment is rounded to
// bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on its way from LDS to the matrix core and the products accumulate in
// fp32 (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate).  The K permutation is the fp32 kernel's: lane half h supplies
// k = 16G + 4h + {0..3} and 16G + 8 + 4h + {0..3} of 16-deep group G for both operands.
__device__ __forceinline__ bf16x8 cvt8(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo[0]; r[1] = (__bf16)lo[1]; r[2] = (__bf16)lo[2]; r[3] = (__bf16)lo[3];
  r[4] = (__bf16)hi[0]; r[5] = (__bf16)hi[1]; r[6] = (__bf16)hi[2]; r[7] = (__bf16)hi[3];
  return r;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x2 pack4_bf16(u32x4 v) {          // 4 fp32 -> 4 bf16 (RNE), channel order kept
  return __builtin_bit_cast(u32x2, __builtin_convertvector(__builtin_bit_cast(f32x4, v), bf16x4_t));
}

// register stages of igemm_f32_kernel's K pipeline per tile class (128x128 / 128x64 and 64x128 / smaller); -D overrides are for A/B builds
#ifndef TG_PF_BIG
#define TG_PF_BIG 1
#endif
#ifndef TG_PF_MID
#define TG_PF_MID 2
#endif
#ifndef TG_PF_BF16_BIG
#define TG_PF_BF16_BIG 1
#endif
#ifndef TG_PF_BF16_MID
#define TG_PF_BF16_MID 2
#endif
#ifndef TG_PF_BF16_SMALL
#define TG_PF_BF16_SMALL 2
#endif
#ifndef TG_INTERLEAVE
#define TG_INTERLEAVE 1               // A/B: one memory instruction behind every MFMA in the K loop of the one-accumulator tiles (see the K loop)
#endif
#ifndef TG_PF_SMALL
#define TG_PF_SMALL 2
#endif

constexpr int BK = 32;    // reduction depth per LDS tile
constexpr int LDT = 36;   // padded LDS row stride (floats)
constexpr int LDH = 80;   // bf16 variant: LDS row stride in BYTES (32 bf16 = 64 B + 16 B pad: the 16 rows of a ds_read_b128 lane group hit 16 different 16-B bank slots)

#ifdef TG_STAMP
// Diagnostic build (never shipped): per-phase cycle sums of the K loop for a few workgroups, wave 0, read back with
// tg_debug_read_stamps.  Stamp values go only to this buffer; no output depends on them.
__device__ unsigned long long tg_stamps[8 * 8];
__device__ unsigned long long tg_block_times[3 * 8192];   // per workgroup: start, end (s_memrealtime, 100 MHz), HW ids
#define STAMP(var)                                                                 \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");    \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define STAMP(var) do {} while (0)
#endif

constexpr int MAX_SUB = 4;
constexpr uint32_t OOB_OFF = 0x80000000u;   // byte offset beyond any (< 2 GiB) tensor: buffer loads return 0, stores are dropped   // sub-problems per launch (the 4 output parities of a stride-2 transposed conv / dgrad)

// kernel-side view of one tg_igemm_desc: tap tables repacked to one dword per tap so that the (block-uniform) tap
// lookup is a scalar s_load_dword — int8/int16 arrays indexed dynamically compile to VECTOR byte loads plus a
// vmcnt wait at the top of every K-tile.
struct SubDesc {
  int32_t h_v, w_v, s_y, s_x, h_out, w_out, ld_out, os_y, os_x, oo_y, oo_x, n_store, n_taps, act;
  float alpha;
  int32_t n_group;
  int32_t taps[TG_MAX_TAPS];     // (tapw << 16) | ((dy & 0xff) << 8) | (dx & 0xff)
};

struct IgemmParams {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  SubDesc d[MAX_SUB];
  int64_t w_sn, w_st;
  int n_img, h_in, w_in, ld_in, c_out;
  int n_sub, M, m_tiles, n_tiles;
  uint32_t in_bytes, w_bytes, out_bytes;
  double* colsum;                 // COLSUM variant: [nseg][c_out] fp64 accumulators (zeroed by the launcher)
  const float* ymul;              // COLSUM variant, optional: out = acc * act'(ymul[same position]) (tg_igemm_actsum_*)
  int ymul_act;
  float ymul_alpha;
  int stat2;                      // COLSUM variant: 2 = the stored value is the accumulator (a gradient dy) and colsum is [nseg][2][c_out]: sums of dy and of
                                  // dy * ymul (a batch norm's backward statistics, tg_igemm_bnbwdstat_*); 1 = the stored value is v = act(acc + bias) and colsum is [nseg][2][c_out]: sums of v and of v*v
                                  // (statistics of a batch norm behind the layer, tg_igemm_bnstat_*)
  int nseg, seg_rows[8];
  // tg_igemm_labels_*: a _conv_cond_concat follows (Model/modle_base.py:239-244) — the output buffer is the concatenated tensor and the workgroups of
  // the last column tile also write channels [lab_c0, lab_c0 + lab_n) = the image's label vector and zeros from there up to ld_out
  const float* lab;
  int lab_n, lab_c0;
  // ---- work units (tg::igemm_schedule, geom.cpp).  A unit is one output tile over a K range; tiles whose K range is cut into ks > 1
  // units leave raw partial accumulators in `ws` and are finished by the fix-up launch (same kernel, FIXUP = true).
  int n_units;                    // grid of the main launch
  int n_fix;                      // grid of the fix-up launch (tiles that were cut), 0: none
  int nfull;                      // pat_len == 0 (one sub-problem): tiles [0, nfull) are whole, tiles [nfull, T) are cut into ks[0] units
  int ks[MAX_SUB];                // units per tile of each sub-problem
  int pat_len;                    // > 0: every tile index owns pat_len consecutive units, unit w of the pattern = (sub, K segment)
  int n_pat_split;                // pattern entries that are partial (per tile index: that many ws slots)
  int n_split_sub;                // sub-problems with ks > 1 (fix-up grid = tiles * n_split_sub)
  int8_t pat_sub[16], pat_k[16], pat_slot[16];      // pat_slot: ws slot of the entry inside its tile index's group, -1: whole tile
  int8_t split_sub[MAX_SUB], first_slot[MAX_SUB];   // the i-th cut sub-problem and the slot of its first K segment
  float* ws;
};


__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  switch (act) {
    case TG_ACT_LRELU: return v > 0.f ? v : alpha * v;
    case TG_ACT_RELU: return v > 0.f ? v : 0.f;
    case TG_ACT_TANH: return tanhf(v);
    case TG_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case TG_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(__expf(v));
    default: return v;
  }
}

// XCD-aware bijective remap (8 XCDs, blocks b and b+8 share one): every XCD works on a contiguous run of logical tiles,
// so the n-tiles that re-read one A tile and neighbouring m-tiles (shared halo rows) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// FIXUP = true: the second launch of a schedule with cut tiles — no K loop: the accumulators are the sum (in K-segment order, so
// deterministic) of the partials the main launch left in p.ws, then the SAME epilogue.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool COLSUM, bool BF16, bool FIXUP = false>
__global__ void __launch_bounds__(256, 2) igemm_f32_kernel(IgemmParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;   // 16-B loads per thread per tile
  // register stages of the K pipeline (see the K loop); bf16 operands have their own depths: a K-tile's MFMAs take a sixteenth of the fp32 time
  constexpr int NBUF = 2;                     // LDS buffers per operand
  constexpr int PF = FIXUP ? 1 : (BF16 ? (BM * BN >= 128 * 128 ? TG_PF_BF16_BIG : (BM * BN >= 128 * 64 ? TG_PF_BF16_MID : TG_PF_BF16_SMALL))
                                       : (BM * BN >= 128 * 128 ? TG_PF_BIG : (BM * BN >= 128 * 64 ? TG_PF_MID : TG_PF_SMALL)));
  __shared__ __attribute__((aligned(16))) float smem[NBUF * BM * LDT + NBUF * BN * LDT + 4 * BM];
  float* As = smem;
  float* Bs = smem + NBUF * BM * LDT;
  int* t_base = reinterpret_cast<int*>(smem + NBUF * BM * LDT + NBUF * BN * LDT);
  int* t_y = t_base + BM;
  int* t_x = t_y + BM;
  int* t_out = t_x + BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef TG_STAMP
  unsigned long long t_entry = 0;
  STAMP(t_entry);
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // Unit -> (sub-problem, tile, K segment).  Several sub-problems (the 9/6/6/4-tap parities of a stride-2 transposed conv) and / or cut
  // tiles: tile-index-major, pattern-minor — every tile index owns pat_len units (sub-problem s contributes ks[s] of them) — and the
  // order inside the pattern rotates every 32 workgroups: workgroups go to the 32 compute units of an XCD round-robin, all resident at
  // once for these small launches, so with a fixed order unit j would get pattern entry j % pat_len every time (measured: the units
  // holding only 9-tap workgroups finish at 190 us, the median at 100); rotating gives every unit the same mix.
  int sub, rem_id, kseg = 0, kcut = 1, slot = -1;
  if constexpr (FIXUP) {
    if (p.pat_len > 0) {
      rem_id = blockIdx.x / p.n_split_sub;
      const int j = blockIdx.x - rem_id * p.n_split_sub;
      sub = p.split_sub[j];
      slot = rem_id * p.n_pat_split + p.first_slot[j];
    } else {
      sub = 0;
      rem_id = p.nfull + blockIdx.x;
      slot = blockIdx.x * p.ks[0];
    }
    kcut = p.ks[sub];
  } else {
    const int lid = xcd_remap(blockIdx.x, p.n_units);
    if (p.pat_len > 0) {
      const int L = p.pat_len;
      rem_id = lid / L;
      const int w = (32 % L == 0) ? (lid + (lid >> 5)) % L : lid - rem_id * L;
      sub = p.pat_sub[w]; kseg = p.pat_k[w]; kcut = p.ks[sub];
      slot = p.pat_slot[w] < 0 ? -1 : rem_id * p.n_pat_split + p.pat_slot[w];
    } else {
      sub = 0;
      if (lid < p.nfull) rem_id = lid;
      else {
        const int v = lid - p.nfull;
        kcut = p.ks[0];
        rem_id = p.nfull + v / kcut;
        kseg = v - (rem_id - p.nfull) * kcut;
        slot = v;
      }
    }
  }
  const SubDesc& d = p.d[sub];
  const int nt = rem_id % p.n_tiles, mt = rem_id / p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;

  if (tid < BM) {
    int m = m0 + tid;
    int base = 0, y0 = -30000, x0 = 0, oo = (int)0x80000000;   // masked row: out-of-range byte offset
    if (m < p.M) {
      int hw = d.h_v * d.w_v;
      int img = m / hw, rem = m - img * hw;
      int vy = rem / d.w_v, vx = rem - vy * d.w_v;
      base = img * p.h_in * p.w_in * p.ld_in;
      y0 = vy * d.s_y;
      x0 = vx * d.s_x;
      oo = ((img * d.h_out + vy * d.os_y + d.oo_y) * d.w_out + vx * d.os_x + d.oo_x) * d.ld_out * 4;   // byte offset
    }
    t_base[tid] = base; t_y[tid] = y0; t_x[tid] = x0; t_out[tid] = oo;
  }
  __syncthreads();

  const int seg = tid & 7, lrow = tid >> 3;
  int abase[AR], ay[AR], ax[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    int r = lrow + 32 * j;
    abase[j] = t_base[r] + seg * 4; ay[j] = t_y[r]; ax[j] = t_x[r];
  }

  // Both operands are read through buffer descriptors with the K-tile position in the SCALAR offset operand:
  //   gathered rows : voffset = byte offset of (pixel, tap) — recomputed once per TAP, not per K-tile — or
  //                   OOB_OFF for an out-of-image tap (the hardware range check returns zeros: no branch, no select);
  //   filter rows   : voffset fixed for the whole kernel;   soffset = (tap, channel chunk) for both.
  // On gfx950 the fp32 MFMA runs on the vector ALUs, so every VALU instruction of the address arithmetic is time the
  // matrix work does not get (stamped build: a wave's load-issue phase is starved for the whole MFMA phase of its SIMD
  // partner).  Per K-tile this loop now issues 8 loads, 8 LDS writes, 16 LDS reads and ~10 scalar instructions.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  uint32_t wvoff[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) wvoff[j] = (uint32_t)(((int64_t)(n0 + lrow + 32 * j) * p.w_sn + seg * 4) * 4);

  u32x4 ra[PF][AR], rb[PF][BR];
  uint32_t avoff[AR];
  const int cchunks = p.ld_in / BK;
  const int nk = d.n_taps * cchunks;
  const int h_in = p.h_in, w_in = p.w_in, ld_in = p.ld_in;
  int w_tap_off = 0;                                    // element offset of the current tap's filter slice

  auto set_tap = [&](int tap) {
    const int tp = d.taps[tap];
    const int dy = (int)(int8_t)(tp >> 8), dx = (int)(int8_t)tp;
    w_tap_off = (tp >> 16) * (int)p.w_st;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int iy = ay[j] + dy, ix = ax[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)h_in && (unsigned)ix < (unsigned)w_in;
      avoff[j] = ok ? (uint32_t)(abase[j] + (iy * w_in + ix) * ld_in) * 4u : OOB_OFF;
    }
  };
  auto gload = [&](int c0, int rs) {                     // rs: register stage (a compile-time constant after unrolling)
    const uint32_t sa = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
    const uint32_t sw = (uint32_t)__builtin_amdgcn_readfirstlane((w_tap_off + c0) * 4);
#pragma unroll
    for (int j = 0; j < AR; ++j) ra[rs][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, avoff[j], sa, 0);
#pragma unroll
    for (int j = 0; j < BR; ++j) rb[rs][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wvoff[j], sw, 0);
  };
  auto sstore = [&](int buf, int rs) {
    if constexpr (BF16) {
      // bf16 LDS images (round 3): the operands are rounded ONCE on the global -> LDS path (v_cvt_pk_bf16_f32, RNE) and a 32-deep row is
      // 64 bytes — half the LDS bytes of the fp32 image, and the fragments below need no conversion (round 2 converted every fragment
      // on its way LDS -> registers: eight packed conversions per operand fragment next to the MFMAs)
      unsigned char* a = reinterpret_cast<unsigned char*>(As) + (buf * BM + lrow) * LDH + seg * 8;
      unsigned char* b = reinterpret_cast<unsigned char*>(Bs) + (buf * BN + lrow) * LDH + seg * 8;
#pragma unroll
      for (int j = 0; j < AR; ++j) *reinterpret_cast<u32x2*>(a + 32 * j * LDH) = pack4_bf16(ra[rs][j]);
#pragma unroll
      for (int j = 0; j < BR; ++j) *reinterpret_cast<u32x2*>(b + 32 * j * LDH) = pack4_bf16(rb[rs][j]);
    } else {
      float* a = As + buf * BM * LDT + lrow * LDT + seg * 4;
      float* b = Bs + buf * BN * LDT + lrow * LDT + seg * 4;
#pragma unroll
      for (int j = 0; j < AR; ++j) *reinterpret_cast<u32x4*>(a + 32 * j * LDT) = ra[rs][j];
#pragma unroll
      for (int j = 0; j < BR; ++j) *reinterpret_cast<u32x4*>(b + 32 * j * LDT) = rb[rs][j];
    }
  };

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int frag = (lane & 31) * LDT + (lane >> 5) * 4;
  constexpr int WS_VEC = MI * NI * 4;                    // float4 pieces of a thread's accumulators; ws piece j of slot s, thread t at ((s * WS_VEC + j) * 256 + t)

  if constexpr (FIXUP) {
    for (int k = 0; k < kcut; ++k) {                     // fixed order: the result does not depend on which unit finished first
      const f32x4* src = reinterpret_cast<const f32x4*>(p.ws) + ((int64_t)(slot + k) * WS_VEC) * 256 + tid;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = src[((mi * NI + ni) * 4 + q) * 256];
            acc[mi][ni][4 * q] += v[0]; acc[mi][ni][4 * q + 1] += v[1]; acc[mi][ni][4 * q + 2] += v[2]; acc[mi][ni][4 * q + 3] += v[3];
          }
    }
  } else {
  // K range of this unit: K-tiles [it0, it1) of the tile's nk (balanced integer cut)
  const int it0 = (int)((int64_t)nk * kseg / kcut), it1 = (int)((int64_t)nk * (kseg + 1) / kcut);

  // Register-staged pipeline over a two-buffer LDS image, PF register stages deep: while tile `it` is multiplied, the loads of tiles
  // it+1 .. it+PF are in flight (tile it+PF is issued at the top of iteration it into the stage that tile it occupied), and tile it+1 is
  // written to the other LDS buffer after the MFMAs (one barrier per K-tile).  PF = 1 is the round-1 scheme: enough for the 128x128 tile
  // (64 MFMAs = 1.7 us per K-tile cover a global-load latency) with two workgroups per CU; a 64x64 / 128x32 tile has 16 MFMAs = 0.43 us
  // per K-tile, and the launches that use them are the under-filled ones (one workgroup on most CUs), so one tile of lookahead leaves the
  // load latency exposed in EVERY K-tile — three stages cover ~1.3 us.
  int tap = it0 / cchunks, c0 = (it0 - tap * cchunks) * BK, buf = 0, issued = it0;
  set_tap(tap);
  auto issue = [&](int rs) {                             // loads of tile `issued` into register stage rs; advances the (tap, channel chunk) cursor
    gload(c0, rs);
    ++issued;
    c0 += BK;
    if (c0 == ld_in) { c0 = 0; ++tap; if (issued < it1) set_tap(tap); }
  };
#ifdef TG_STAMP
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, acc_load = 0, acc_mfma = 0, acc_store = 0, acc_bar = 0, t_begin = 0;
#endif
#pragma unroll
  for (int rs = 0; rs < PF; ++rs)
    if (it0 + rs < it1) issue(rs);
  sstore(0, 0);
  __syncthreads();

#ifdef TG_STAMP
  STAMP(t_begin);
#endif
  // one K-tile: the MFMAs of the tile in LDS buffer `b`
  auto compute = [&](int b) {
    if constexpr (BF16) {
      // lane (row r = lane & 31, half h = lane >> 5) supplies k = 16 G + 8 h + (0..7) of 16-deep group G for BOTH operands: one 16-byte read
      const unsigned char* Ab = reinterpret_cast<const unsigned char*>(As) + (b * BM + wm0 + (lane & 31)) * LDH + (lane >> 5) * 16;
      const unsigned char* Bb = reinterpret_cast<const unsigned char*>(Bs) + (b * BN + wn0 + (lane & 31)) * LDH + (lane >> 5) * 16;
#pragma unroll
      for (int G = 0; G < BK / 16; ++G) {
        bf16x8 fa[MI], fb[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(Ab + mi * 32 * LDH + G * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const bf16x8*>(Bb + ni * 32 * LDH + G * 32);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            if (COLSUM) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
            else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
      }
    } else {
      const float* A = As + b * BM * LDT + wm0 * LDT + frag;
      const float* B = Bs + b * BN * LDT + wn0 * LDT + frag;
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        f32x4 fa[MI], fb[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const f32x4*>(A + mi * 32 * LDT + g * 8);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const f32x4*>(B + ni * 32 * LDT + g * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              if (COLSUM) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][s], fb[ni][s], acc[mi][ni], 0, 0, 0);   // D[m][n]: lane = channel
              else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni][s], fa[mi][s], acc[mi][ni], 0, 0, 0);        // D[n][m]: lane = pixel
      }
    }
  };
  int it = it0;
  // steady state, PF iterations at a time, each issuing one tile UNCONDITIONALLY: straight-line code, so the wait in front of the LDS
  // write of stage rs + 1 is a counted one (the PF - 1 younger tiles stay in flight) — with a branch around the issue the compiler has to
  // assume the youngest loads are the ones it needs and drains the queue every iteration
  if constexpr (TG_INTERLEAVE && MI == 1 && NI == 1 && !BF16 && PF >= 2 && 2 * (AR + BR) <= 10) {
    // Tiles with ONE accumulator per wave (64x64, 128x32, 32x128): the 16 MFMAs of a K-tile form one dependent chain, and a wave issues in
    // order — while MFMA k runs (64 cycles) the wave sits at MFMA k+1, so anything placed before or behind the chain is time the matrix pipe
    // idles when no second wave shares the SIMD (launches with at most one workgroup per CU: profiles/r04_stamp_small.txt).  Here every
    // memory instruction of the K-tile — six fragment reads, the AR + BR global loads of tile it+PF, the AR + BR LDS writes of tile it+1 —
    // sits in its own slot BEHIND one MFMA (fenced, so that the compiler keeps it there) and issues in that MFMA's shadow
    // (profiles/r04_interleave_ab.txt: -6.5 % over the generic launches of the step, -15 ... -18 % where one workgroup owns a CU).
    constexpr int NL = AR + BR;
    // one K-tile; rs (register stage of the tile being loaded) is a constant at every call site.  do_load / do_write: the steady state passes
    // true (straight-line code, counted waits), the drain what is left to load / to write
    auto ktile = [&](const int rs, const bool do_load, const bool do_write) __attribute__((always_inline)) {
      const int ws = (rs + 1) % PF;                       // register stage of tile it+1 (its loads are a whole iteration old)
      const uint32_t sa = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
      const uint32_t sw = (uint32_t)__builtin_amdgcn_readfirstlane((w_tap_off + c0) * 4);
      const float* A = As + buf * BM * LDT + wm0 * LDT + frag;
      const float* B = Bs + buf * BN * LDT + wn0 * LDT + frag;
      float* wa = As + (buf ^ 1) * BM * LDT + lrow * LDT + seg * 4;
      float* wb = Bs + (buf ^ 1) * BN * LDT + lrow * LDT + seg * 4;
      f32x4 fa[2], fb[2];
      fa[0] = *reinterpret_cast<const f32x4*>(A);
      fb[0] = *reinterpret_cast<const f32x4*>(B);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int g = k >> 2, sft = k & 3, cur = g & 1;
        if (COLSUM) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][sft], fb[cur][sft], acc[0][0], 0, 0, 0);
        else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][sft], fa[cur][sft], acc[0][0], 0, 0, 0);
        // slot k: k = 4g, 4g+1 (g < 3): the fragments of group g+1; the other ten slots: NL loads, then NL LDS writes
        if (sft == 0 && g < 3) fa[cur ^ 1] = *reinterpret_cast<const f32x4*>(A + (g + 1) * 8);
        else if (sft == 1 && g < 3) fb[cur ^ 1] = *reinterpret_cast<const f32x4*>(B + (g + 1) * 8);
        else {
          const int j = (g < 3) ? 2 * g + (sft - 2) : 6 + sft;           // 0 .. 9 over the free slots (2,3,6,7,10,11,12..15)
          if (j < NL) {
            if (do_load) {
              if (j < AR) ra[rs][j < AR ? j : 0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, avoff[j < AR ? j : 0], sa, 0);
              else rb[rs][j >= AR ? j - AR : 0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wvoff[j >= AR ? j - AR : 0], sw, 0);
            }
          } else if (j < 2 * NL) {
            const int q = j - NL;
            if (do_write) {
              if (q < AR) *reinterpret_cast<u32x4*>(wa + 32 * (q < AR ? q : 0) * LDT) = ra[ws][q < AR ? q : 0];
              else *reinterpret_cast<u32x4*>(wb + 32 * (q >= AR ? q - AR : 0) * LDT) = rb[ws][q >= AR ? q - AR : 0];
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (do_load) {                                       // the cursor of the tile just issued (what issue() does behind its loads)
        ++issued;
        c0 += BK;
        if (c0 == ld_in) { c0 = 0; ++tap; if (issued < it1) set_tap(tap); }
      }
      __syncthreads();
      buf ^= 1;
    };
    while (issued + PF <= it1) {
#pragma unroll
      for (int rs = 0; rs < PF; ++rs) ktile(rs, true, true);
      it += PF;
    }
    for (; it < it1; it += PF) {                           // drain (and K ranges shorter than the pipeline)
#pragma unroll
      for (int rs = 0; rs < PF; ++rs)
        if (it + rs < it1) ktile(rs, issued < it1, it + rs + 1 < it1);
    }
  }
  while (issued + PF <= it1) {
#pragma unroll
    for (int rs = 0; rs < PF; ++rs) {
      STAMP(ts0);
      issue(rs);                                         // stage rs went to LDS in the previous iteration (or in the prologue)
      STAMP(ts1);
      compute(buf);
      STAMP(ts2);
      sstore(buf ^ 1, (rs + 1) % PF);
      STAMP(ts3);
      __syncthreads();
      STAMP(ts4);
#ifdef TG_STAMP
      acc_load += ts1 - ts0; acc_mfma += ts2 - ts1; acc_store += ts3 - ts2; acc_bar += ts4 - ts3;
#endif
      buf ^= 1;
    }
    it += PF;
  }
  // drain (and K ranges shorter than the pipeline): fewer than 2 PF iterations; the stage loop must unroll completely — its index names
  // registers — hence guards, not breaks
  for (; it < it1; it += PF) {
#pragma unroll
    for (int rs = 0; rs < PF; ++rs) {
      if (it + rs < it1) {
        if (issued < it1) issue(rs);
        compute(buf);
        if (it + rs + 1 < it1) sstore(buf ^ 1, (rs + 1) % PF);
        __syncthreads();
        buf ^= 1;
      }
    }
  }
#ifdef TG_STAMP
  {
    unsigned long long t_end;
    STAMP(t_end);
    const int sslot = blockIdx.x == 0 ? 0 : (blockIdx.x == 7 ? 1 : (blockIdx.x == 300 ? 2 : (blockIdx.x == 700 ? 3 : (blockIdx.x == 1500 ? 4 : -1))));
    if (sslot >= 0 && tid == 0) {
      tg_stamps[sslot * 8 + 0] = acc_load; tg_stamps[sslot * 8 + 1] = acc_mfma; tg_stamps[sslot * 8 + 2] = acc_store;
      tg_stamps[sslot * 8 + 3] = acc_bar; tg_stamps[sslot * 8 + 4] = t_end - t_begin; tg_stamps[sslot * 8 + 5] = it1 - it0;
      tg_stamps[sslot * 8 + 6] = t_begin - t_entry;
    }
  }
#endif
  if (slot >= 0) {
    // a cut tile: this unit's accumulators go to the workspace as they lie in the registers (16 B per lane, 1 KB per wave-instruction) and
    // the fix-up launch adds the tile's K segments up and runs the epilogue
    f32x4* dst = reinterpret_cast<f32x4*>(p.ws) + ((int64_t)slot * WS_VEC) * 256 + tid;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
          dst[((mi * NI + ni) * 4 + q) * 256] = v;
        }
    return;
  }
  }  // !FIXUP
#ifdef TG_STAMP
  unsigned long long t_epi0 = 0;
  STAMP(t_epi0);
#endif

  // epilogue.  The MFMAs were issued with the FILTER as the row operand, so a lane holds ONE output pixel (lane & 31)
  // and its 16 registers hold channels 8q + 4h + (0..3): four consecutive channels of that pixel are adjacent in NHWC
  // memory -> 16 dwordx4 stores per lane instead of 64 dword stores (measured: the store tail of a 128x128 tile
  // drops from ~36k to ~10k cycles).  +bias, activation; masked rows carry an out-of-range byte offset in the row
  // table, masked channel groups get one here, and the buffer range check drops those stores (no branches).
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const int half = lane >> 5, col = lane & 31;
  const uint32_t* t_ob = reinterpret_cast<const uint32_t*>(t_out);
  if (COLSUM) {
    // Mean-only-BN variants.  The pixel operand was the MFMA row operand, so a lane holds ONE output channel and its 16 registers
    // hold rows: storing from that layout means 64 dword stores per lane (and, for tg_igemm_actsum_*, 64 dword loads of the
    // activation).  The tile goes through LDS instead (the operand tiles are dead after the K loop) and leaves it row-wise:
    // every thread moves 16-B pieces of pixel rows — coalesced stores, coalesced activation loads — and the column sums are taken
    // from LDS by (column, row-range) threads.  A tile may straddle ONE application boundary (every segment has at least BM rows:
    // launcher check): rows below `bnd` (tile-local) belong to segment `seg`, the others to seg + 1.
    constexpr int TLD = BN + 4;                           // LDS row stride of the transposed tile (floats)
    static_assert(BM * TLD <= 2 * BM * LDT + 2 * BN * LDT, "the output tile must fit into the operand tiles' LDS");
    float* tile = smem;
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ymul ? p.ymul : p.out), 0, p.out_bytes, 0x00020000);
    int seg = 0, acc_rows = p.seg_rows[0];
    while (seg < p.nseg - 1 && m0 >= acc_rows) acc_rows += p.seg_rows[++seg];
    const int bnd = acc_rows - m0;                        // >= 1; >= BM when the tile lies inside one segment
    const bool two = bnd < BM && seg + 1 < p.nseg;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TLD + wn0 + ni * 32 + col] = acc[mi][ni][r];
    __syncthreads();
    constexpr int G4 = BN / 4;                            // 16-B pieces per tile row
    const bool bst = p.stat2 == 2;                        // tg_igemm_bnbwdstat_*: sums of dy and dy * x, x = p.ymul read at the output's addresses
    const bool ym = p.ymul != nullptr && !bst;
    float bs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, bq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int i = tid; i < BM * G4; i += 256) {
      const int rl = i / G4, cg = i - rl * G4;
      const uint32_t ro = t_ob[rl];
      const int n = n0 + cg * 4;
      const uint32_t off = ((ro & OOB_OFF) || n >= d.n_store) ? OOB_OFF : ro + (uint32_t)n * 4u;
      const float4 tv = *reinterpret_cast<const float4*>(tile + rl * TLD + cg * 4);
      float va[4] = {tv.x, tv.y, tv.z, tv.w};
      if (bst) {      // this thread's pieces all belong to ONE 4-column group (256 % G4 == 0): the sums stay in registers until the tile is done
        const u32x4 xb = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, off, 0, 0);          // masked positions read 0 (and hold acc = 0)
        const uint32_t x0 = xb.x, x1 = xb.y, x2 = xb.z, x3 = xb.w;       // (a bit_cast straight from a vector element reads element 0: take scalars first)
        const float xs[4] = {__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1), __builtin_bit_cast(float, x2), __builtin_bit_cast(float, x3)};
        const float m0_ = rl < bnd ? 1.f : 0.f, m1_ = 1.f - m0_;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pr = va[e] * xs[e];
          bs[0][e] += m0_ * va[e]; bs[1][e] += m1_ * va[e];
          bq[0][e] += m0_ * pr;    bq[1][e] += m1_ * pr;
        }
      } else if (p.stat2) {  // batch norm behind the layer: bias + activation here, statistics of the result; rows beyond M must stay exact zeros
        const bool live = !(ro & OOB_OFF);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = va[e] + ((p.bias != nullptr && n + e < d.n_store) ? p.bias[n + e] : 0.f);
          va[e] = live ? apply_act(t, d.act, d.alpha) : 0.f;
        }
        *reinterpret_cast<float4*>(tile + rl * TLD + cg * 4) = make_float4(va[0], va[1], va[2], va[3]);
      }
      if (ym) {      // input gradient times the activation derivative of the layer that produced this conv's input
        const u32x4 yb = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, off, 0, 0);          // masked positions read 0
        const uint32_t y0 = yb.x, y1 = yb.y, y2 = yb.z, y3 = yb.w;
        va[0] *= tgd::act_grad(__builtin_bit_cast(float, y0), p.ymul_act, p.ymul_alpha);
        va[1] *= tgd::act_grad(__builtin_bit_cast(float, y1), p.ymul_act, p.ymul_alpha);
        va[2] *= tgd::act_grad(__builtin_bit_cast(float, y2), p.ymul_act, p.ymul_alpha);
        va[3] *= tgd::act_grad(__builtin_bit_cast(float, y3), p.ymul_act, p.ymul_alpha);
        *reinterpret_cast<float4*>(tile + rl * TLD + cg * 4) = make_float4(va[0], va[1], va[2], va[3]);
      }
      const u32x4 pk = {__builtin_bit_cast(uint32_t, va[0]), __builtin_bit_cast(uint32_t, va[1]), __builtin_bit_cast(uint32_t, va[2]),
                        __builtin_bit_cast(uint32_t, va[3])};
      __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_o, off, 0, 0);
    }
    if (bst) {
      // the tile is dead: the per-thread sums go through its LDS as [sum kind][e][thread], column n0 + 4 cg + e = threads cg, cg + G4, ...
      __syncthreads();
      static_assert(16 * 256 <= BM * TLD, "the per-thread sums must fit into the tile");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tile[(0 * 4 + e) * 256 + tid] = bs[0][e];
        tile[(1 * 4 + e) * 256 + tid] = bs[1][e];
        tile[(2 * 4 + e) * 256 + tid] = bq[0][e];
        tile[(3 * 4 + e) * 256 + tid] = bq[1][e];
      }
      __syncthreads();
      if (tid < BN) {
        const int n = n0 + tid, cg = tid >> 2, e = tid & 3;
        float s1 = 0.f, s2 = 0.f, q1 = 0.f, q2 = 0.f;
        for (int t2 = cg; t2 < 256; t2 += G4) {
          s1 += tile[(0 * 4 + e) * 256 + t2]; s2 += tile[(1 * 4 + e) * 256 + t2];
          q1 += tile[(2 * 4 + e) * 256 + t2]; q2 += tile[(3 * 4 + e) * 256 + t2];
        }
        if (n < d.n_store) {
          atomicAdd(p.colsum + ((int64_t)seg * 2) * p.c_out + n, (double)s1);
          atomicAdd(p.colsum + ((int64_t)seg * 2 + 1) * p.c_out + n, (double)q1);
          if (two) {
            atomicAdd(p.colsum + ((int64_t)(seg + 1) * 2) * p.c_out + n, (double)s2);
            atomicAdd(p.colsum + ((int64_t)(seg + 1) * 2 + 1) * p.c_out + n, (double)q2);
          }
        }
      }
      return;
    }
    if (ym || p.stat2) __syncthreads();
    // column sums: thread (column c, part q) adds rows q, q + PARTS, ... of its column (masked rows hold exact zeros)
    constexpr int PARTS = 256 / BN;
    float* red = tile + BM * TLD;                         // [4][PARTS][BN] partial sums (two segments; stat2: also of the squares), behind the tile
    static_assert(BM * TLD + 4 * PARTS * BN <= 2 * BM * LDT + 2 * BN * LDT, "partial sums must fit as well");
    {
      const int c = tid % BN, q = tid / BN;
      float s1 = 0.f, s2 = 0.f, q1 = 0.f, q2 = 0.f;
      for (int rl = q; rl < BM; rl += PARTS) {
        const float v = tile[rl * TLD + c];
        if (rl < bnd) { s1 += v; q1 += v * v; } else { s2 += v; q2 += v * v; }
      }
      red[q * BN + c] = s1;
      red[(PARTS + q) * BN + c] = s2;
      red[(2 * PARTS + q) * BN + c] = q1;
      red[(3 * PARTS + q) * BN + c] = q2;
    }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      float s1 = 0.f, s2 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
      for (int q = 0; q < PARTS; ++q) {
        s1 += red[q * BN + tid]; s2 += red[(PARTS + q) * BN + tid];
        q1 += red[(2 * PARTS + q) * BN + tid]; q2 += red[(3 * PARTS + q) * BN + tid];
      }
      if (n < d.n_store) {
        if (p.stat2) {
          atomicAdd(p.colsum + ((int64_t)seg * 2) * p.c_out + n, (double)s1);
          atomicAdd(p.colsum + ((int64_t)seg * 2 + 1) * p.c_out + n, (double)q1);
          if (two) {
            atomicAdd(p.colsum + ((int64_t)(seg + 1) * 2) * p.c_out + n, (double)s2);
            atomicAdd(p.colsum + ((int64_t)(seg + 1) * 2 + 1) * p.c_out + n, (double)q2);
          }
        } else {
          atomicAdd(p.colsum + (int64_t)seg * p.c_out + n, (double)s1);
          if (two) atomicAdd(p.colsum + (int64_t)(seg + 1) * p.c_out + n, (double)s2);
        }
      }
    }
    return;
  }
  const int ng = d.n_group;                                    // > 0: column -> (output-pixel parity, channel), see tg_igemm_desc
  const bool vec_ok = (d.n_store & 3) == 0 && (d.ld_out & 3) == 0 && (ng & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t ro = t_ob[wm0 + mi * 32 + col];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn0 + ni * 32 + 8 * q + 4 * half;
        int ch[4];
        uint32_t goff[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ng > 0) {
            const int g = (n + e) / ng;
            ch[e] = n + e - g * ng;
            goff[e] = (uint32_t)(((g / d.os_x) * d.w_out + (g % d.os_x)) * d.ld_out) * 4u;
            if (g >= d.os_x * d.os_y) ch[e] = d.n_store;          // padding columns behind the last group: never stored
          } else {
            ch[e] = n + e;
            goff[e] = 0;
          }
        }
        float va[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[mi][ni][4 * q + e];
          if (p.bias != nullptr && ch[e] < d.n_store) t += p.bias[ch[e]];
          va[e] = apply_act(t, d.act, d.alpha);
        }
        if (vec_ok) {
          const uint32_t off = ((ro & OOB_OFF) || ch[0] >= d.n_store) ? OOB_OFF : ro + goff[0] + (uint32_t)ch[0] * 4u;
          const u32x4 pk = {__builtin_bit_cast(uint32_t, va[0]), __builtin_bit_cast(uint32_t, va[1]), __builtin_bit_cast(uint32_t, va[2]),
                            __builtin_bit_cast(uint32_t, va[3])};
          __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_o, off, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t off = ((ro & OOB_OFF) || ch[e] >= d.n_store) ? OOB_OFF : ro + goff[e] + (uint32_t)ch[e] * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, va[e]), rsrc_o, off, 0, 0);
          }
        }
      }
    }
  }
  if (p.lab != nullptr && nt == p.n_tiles - 1 && wn0 == 0) {
    // the label channels and the channel padding of the concatenated tensor: lane = output pixel, the two lane halves alternate 16-byte groups
    const int hw = d.h_v * d.w_v;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rl = wm0 + mi * 32 + col;
      const uint32_t ro = t_ob[rl];
      const bool live = !(ro & OOB_OFF);
      const float* lrow = p.lab + (int64_t)((m0 + rl) / hw) * p.lab_n;
      for (int c4 = p.lab_c0 + 4 * half; c4 < d.ld_out; c4 += 8) {
        float va[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ch = c4 + e - p.lab_c0;
          va[e] = (live && ch < p.lab_n) ? lrow[ch] : 0.f;
        }
        const u32x4 pk = {__builtin_bit_cast(uint32_t, va[0]), __builtin_bit_cast(uint32_t, va[1]), __builtin_bit_cast(uint32_t, va[2]),
                          __builtin_bit_cast(uint32_t, va[3])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_o, live ? ro + (uint32_t)c4 * 4u : OOB_OFF, 0, 0);
      }
    }
  }
#ifdef TG_STAMP
  {
    unsigned long long t_epi1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(t_epi1);
    const int sslot = blockIdx.x == 0 ? 0 : (blockIdx.x == 7 ? 1 : (blockIdx.x == 300 ? 2 : (blockIdx.x == 700 ? 3 : (blockIdx.x == 1500 ? 4 : -1))));
    if (sslot >= 0 && tid == 0) tg_stamps[sslot * 8 + 7] = t_epi1 - t_epi0;
    if (tid == 0 && blockIdx.x < 8192) {
      tg_block_times[3 * blockIdx.x] = rt0;
      tg_block_times[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
      tg_block_times[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                           (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// filter gradient: rows = reduction channel c of `in`, cols = n of `dout`, reduction over pixels.
// LDS tiles are the natural NHWC rows [32 px][CT] / [32 px][NT]; the MFMA k index of step s, lane half h
// is pixel s+16h, read with conflict-free ds_read_b32 (32 consecutive dwords per lane half).
// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const float* in;
  const float* dout;
  float* slab;
  tg_igemm_desc d;
  int M, n_split, px_per_split, c_tiles, n_tiles;
  uint32_t in_bytes, dout_bytes;
};

constexpr uint32_t WG_INVALID = 0x80000000u;   // byte offset beyond any (< 2 GiB) operand: the buffer range check yields zeros

template <int CT, int NT, int WAVES_C, int WAVES_N, int WAVES_K, bool BF16>
__global__ void __launch_bounds__(256, 2) wgrad_f32_kernel(WgradParams p) {
  static_assert(WAVES_C * WAVES_N * WAVES_K == 4, "4 waves");
  constexpr int WC = CT / WAVES_C, WN = NT / WAVES_N, MI = WC / 32, NI = WN / 32;
  constexpr int AR = CT / 32, BR = NT / 32;            // 16-B loads per thread per tile
  constexpr int ASEG = CT / 4, BSEG = NT / 4;          // 16-B segments per pixel row
  constexpr int TILE = BK * (CT + NT);
  // bf16 operands, 128 x 128 tiles (the classifier's large layers): both LDS images hold bf16 in 256-byte pixel rows — converted once on
  // the global -> LDS path — and the pixel-major (K-major) fragments are read with the transposing ds_read_b64_tr_b16: two reads per
  // 32x32x16 fragment instead of eight ds_read_b32 + four conversions, half the LDS bytes.  16-byte chunk index XOR-swizzled by
  // ((row & 3) << 2) | ((row >> 2) & 3): row stores and transposed reads are then both conflict-free.  Other tile shapes keep fp32 images.
  constexpr bool TR = BF16 && CT == 128 && NT == 128;
  constexpr int RED = (WAVES_K > 1) ? (WAVES_K - 1) * 64 * 16 * MI * NI * WAVES_C * WAVES_N : 0;
  constexpr int SM = (2 * TILE > RED ? 2 * TILE : RED) + 8 * BK;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  int* tbl = reinterpret_cast<int*>(smem + (2 * TILE > RED ? 2 * TILE : RED));   // [4 slots][2][BK]: in_off, out_off of pixel tile t in slot t & 3

  const tg_igemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: consecutive logical workgroups (the tiles and the 9 taps of one pixel split, which stream the same
  // `dout` rows and neighbouring `in` rows at the same pace) run on ONE XCD and meet in its L2
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = b % p.n_tiles; b /= p.n_tiles;
  const int ct = b % p.c_tiles; b /= p.c_tiles;
  const int tap = b % d.n_taps;
  const int split = b / d.n_taps;
  const int c0 = ct * CT, n0 = nt * NT;
  const int p_begin = split * p.px_per_split;
  const int p_end = min(p.M, p_begin + p.px_per_split);
  const int nk = (p_end > p_begin) ? (p_end - p_begin + BK - 1) / BK : 0;
  const int dy = d.dy[tap], dx = d.dx[tap];

  // threads < BK walk their pixel (img, vy, vx) forward by BK per K-tile instead of dividing: the table fill sits on
  // wave 0's path to the barrier, two integer divisions per tile cost ~10 % of the MFMA time.
  int f_img = 0, f_vy = 0, f_vx = 0, f_px = p_begin + tid;
  if (tid < BK) {
    const int hw = d.h_v * d.w_v;
    f_img = f_px / hw;
    const int rem = f_px - f_img * hw;
    f_vy = rem / d.w_v;
    f_vx = rem - f_vy * d.w_v;
  }
  auto fill_tbl = [&](int it) {          // threads < BK: offsets of pixel tile `it`; MUST be called for it = 0, 1, 2, ... in order
    uint32_t io = WG_INVALID, oo = WG_INVALID;
    if (f_px < p_end) {
      const int iy = f_vy * d.s_y + dy, ix = f_vx * d.s_x + dx;
      if ((unsigned)iy < (unsigned)d.h_in && (unsigned)ix < (unsigned)d.w_in)
        io = (uint32_t)(((f_img * d.h_in + iy) * d.w_in + ix) * d.ld_in) * 4u;
      oo = (uint32_t)(((f_img * d.h_out + f_vy * d.os_y + d.oo_y) * d.w_out + f_vx * d.os_x + d.oo_x) * d.ld_out) * 4u;
    }
    tbl[(it & 3) * 2 * BK + tid] = (int)io;
    tbl[(it & 3) * 2 * BK + BK + tid] = (int)oo;
    f_px += BK;
    f_vx += BK;
    while (f_vx >= d.w_v) { f_vx -= d.w_v; ++f_vy; }
    while (f_vy >= d.h_v) { f_vy -= d.h_v; ++f_img; }
  };

  const int aseg = tid % ASEG, arow = tid / ASEG;      // rows advance by 256/ASEG per pass
  const int bseg = tid % BSEG, brow = tid / BSEG;
  // both operands through buffer descriptors: masked pixels carry WG_INVALID and read zeros without a branch, so the
  // loads stay in flight behind the MFMAs (see igemm_f32_kernel)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, p.dout_bytes, 0x00020000);
  const uint32_t a_add = (uint32_t)(c0 + aseg * 4) * 4u, b_add = (uint32_t)(n0 + bseg * 4) * 4u;
  u32x4 ra[AR], rb[BR];
  auto gload = [&](int it) {
    const int* ti = tbl + (it & 3) * 2 * BK;
#pragma unroll
    for (int j = 0; j < AR; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (uint32_t)ti[arow + j * (256 / ASEG)] + a_add, 0, 0);
#pragma unroll
    for (int j = 0; j < BR; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_do, (uint32_t)ti[BK + brow + j * (256 / BSEG)] + b_add, 0, 0);
  };
  auto tr_off = [](int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };
  auto sstore = [&](int buf) {
    float* a = smem + buf * TILE;
    float* bb = a + BK * CT;
    if constexpr (TR) {
      unsigned char* a8 = reinterpret_cast<unsigned char*>(a);
      unsigned char* b8 = a8 + BK * 256;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const int row = arow + j * (256 / ASEG);
        *reinterpret_cast<u32x2*>(a8 + tr_off(row, aseg >> 1) + 8 * (aseg & 1)) = pack4_bf16(ra[j]);
      }
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        const int row = brow + j * (256 / BSEG);
        *reinterpret_cast<u32x2*>(b8 + tr_off(row, bseg >> 1) + 8 * (bseg & 1)) = pack4_bf16(rb[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < AR; ++j) *reinterpret_cast<u32x4*>(a + (arow + j * (256 / ASEG)) * CT + aseg * 4) = ra[j];
#pragma unroll
      for (int j = 0; j < BR; ++j) *reinterpret_cast<u32x4*>(bb + (brow + j * (256 / BSEG)) * NT + bseg * 4) = rb[j];
    }
  };

  const int wk = wave % WAVES_K;
  const int wcn = wave / WAVES_K;
  const int wc0 = (wcn / WAVES_N) * WC, wn0 = (wcn % WAVES_N) * WN;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // same two-deep software pipeline as igemm_f32_kernel: registers hold tile it+1, LDS buffer `buf` tile it.  The offset
  // table has 4 slots: iteration it reads slot (it+2)&3 for its loads and fills slot (it+3)&3 (last read two iterations
  // ago) before the barrier that publishes it.
  const int half = lane >> 5, col = lane & 31;
  int buf = 0;
  if (nk > 0) {
    if (tid < BK) { fill_tbl(0); if (nk > 1) fill_tbl(1); if (nk > 2) fill_tbl(2); }
    __syncthreads();
    gload(0);
    sstore(0);
    if (nk > 1) gload(1);
    __syncthreads();
  }
  for (int it = 0; it < nk; ++it) {
    if (it + 1 < nk) sstore(buf ^ 1);
    if (it + 2 < nk) gload(it + 2);
    if (tid < BK && it + 3 < nk) fill_tbl(it + 3);
    const float* A = smem + buf * TILE + (16 * half) * CT + wc0 + col;
    const float* B = smem + buf * TILE + BK * CT + (16 * half) * NT + wn0 + col;
    if constexpr (TR) {
      // fragment of lane (column i = lane & 31, k-group h = lane >> 5): pixels 16G + 8h + {0..7} of channel (tile column) i.  A transposed
      // read serves 16 lanes with a block of 4 pixel rows x 16 channels: lane 4q + p of the group addresses row q, channels 4p..4p+3 and
      // receives the four rows of channel (lane & 15); two blocks (rows +0..3, +4..7) make the eight k values.
      const unsigned char* a8 = reinterpret_cast<const unsigned char*>(smem + buf * TILE);
      const unsigned char* b8 = a8 + BK * 256;
      const int q = (lane >> 2) & 3, cq = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);        // row within the block, first channel this lane addresses
#pragma unroll
      for (int G = wk; G < BK / 16; G += WAVES_K) {
        bf16x8 a[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int cs = wc0 + mi * 32 + cq;
          const int r0 = 16 * G + 8 * half + q;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a8 + tr_off(r0, cs >> 3) + 8 * ((cs >> 2) & 1)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a8 + tr_off(r0 + 4, cs >> 3) + 8 * ((cs >> 2) & 1)));
          a[mi] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int cs = wn0 + ni * 32 + cq;
          const int r0 = 16 * G + 8 * half + q;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b8 + tr_off(r0, cs >> 3) + 8 * ((cs >> 2) & 1)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b8 + tr_off(r0 + 4, cs >> 3) + 8 * ((cs >> 2) & 1)));
          bv[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bv[ni], acc[mi][ni], 0, 0, 0);
      }
    } else if constexpr (BF16) {
      // lane half h supplies pixels 16G + 8h + {0..7} of 16-pixel group G (eight conflict-free ds_read_b32 per fragment)
      const float* Ab = smem + buf * TILE + (8 * half) * CT + wc0 + col;
      const float* Bb = smem + buf * TILE + BK * CT + (8 * half) * NT + wn0 + col;
#pragma unroll
      for (int G = wk; G < BK / 16; G += WAVES_K) {
        bf16x8 a[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < 8; ++j) a[mi][j] = (__bf16)Ab[(16 * G + j) * CT + mi * 32];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int j = 0; j < 8; ++j) bv[ni][j] = (__bf16)Bb[(16 * G + j) * NT + ni * 32];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bv[ni], acc[mi][ni], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int s = wk; s < 16; s += WAVES_K) {
      float a[MI], bv[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = A[s * CT + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni] = B[s * NT + ni * 32];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
    }
    __syncthreads();
    buf ^= 1;
  }

  if (WAVES_K > 1) {   // cross-wave reduction of the split reduction (small-channel layers)
    float* red = smem;
    __syncthreads();
    if (wk > 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[((((wk - 1) * WAVES_C * WAVES_N + wcn) * MI * NI + mi * NI + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int k = 1; k < WAVES_K; ++k)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[mi][ni][r] += red[((((k - 1) * WAVES_C * WAVES_N + wcn) * MI * NI + mi * NI + ni) * 16 + r) * 64 + lane];
    }
  }
  if (wk == 0) {
    float* out = p.slab + ((int64_t)(split * d.n_taps + tap) * d.ld_in) * d.c_out;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int c = c0 + wc0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          int n = n0 + wn0 + ni * 32 + col;
          if (c < d.ld_in && n < d.c_out) out[(int64_t)c * d.c_out + n] = acc[mi][ni][r];
        }
  }
}

int check_desc(const tg_igemm_desc* d) {
  TG_REQUIRE(d != nullptr, "igemm: null descriptor");
  TG_REQUIRE(d->ld_in > 0 && d->ld_in % 32 == 0, "igemm: ld_in=%d must be a positive multiple of 32", d->ld_in);
  TG_REQUIRE(d->c_out > 0 && d->c_out % 32 == 0, "igemm: c_out=%d must be a positive multiple of 32", d->c_out);
  TG_REQUIRE(d->n_taps >= 1 && d->n_taps <= TG_MAX_TAPS, "igemm: n_taps=%d out of range", d->n_taps);
  TG_REQUIRE(d->n_img > 0 && d->h_v > 0 && d->w_v > 0 && d->h_in > 0 && d->w_in > 0, "igemm: empty geometry");
  TG_REQUIRE(d->n_store >= 0 && d->n_store <= d->c_out && d->n_store <= d->ld_out, "igemm: n_store=%d vs c_out=%d ld_out=%d",
             d->n_store, d->c_out, d->ld_out);
  TG_REQUIRE(d->n_group >= 0 && (d->n_group == 0 || (d->n_store <= d->n_group && d->n_group * d->os_y * d->os_x <= d->c_out)),
             "igemm: n_group=%d vs n_store=%d c_out=%d os=%dx%d", d->n_group, d->n_store, d->c_out, d->os_y, d->os_x);
  const int gy_ = d->n_group ? d->os_y - 1 : 0, gx_ = d->n_group ? d->os_x - 1 : 0;
  TG_REQUIRE((d->h_v - 1) * d->os_y + d->oo_y + gy_ < d->h_out && (d->w_v - 1) * d->os_x + d->oo_x + gx_ < d->w_out && d->oo_y >= 0 && d->oo_x >= 0,
             "igemm: virtual grid %dx%d (stride %d,%d offset %d,%d) exceeds output %dx%d", d->h_v, d->w_v, d->os_y, d->os_x,
             d->oo_y, d->oo_x, d->h_out, d->w_out);
  TG_REQUIRE((int64_t)d->n_img * d->h_in * d->w_in * d->ld_in < (1LL << 31) && (int64_t)d->n_img * d->h_out * d->w_out * d->ld_out < (1LL << 31),
             "igemm: tensor exceeds 2^31 elements");
  return TG_OK;
}

}  // namespace

template <int BM, int BN, int WM_, int WN_>
static void launch_igemm(IgemmParams& p, hipStream_t s, bool bf16) {
  const dim3 grid(p.n_units);               // one workgroup per work unit (tg::igemm_schedule); the last column tile may overhang: filter rows >= c_out read zeros / unused data, stores are masked by n_store
  if (bf16) {
    if (p.colsum) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM_, WN_, true, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM_, WN_, false, true>), grid, dim3(256), 0, s, p);
  } else {
    if (p.colsum) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM_, WN_, true, false>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM_, WN_, false, false>), grid, dim3(256), 0, s, p);
  }
  if (p.n_fix > 0) {                        // tiles that were cut along K: add their partial sums up, then the usual epilogue (operand type plays no part)
    if (p.colsum) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM_, WN_, true, false, true>), dim3(p.n_fix), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM_, WN_, false, false, true>), dim3(p.n_fix), dim3(256), 0, s, p);
  }
}

// the head / tail cut of a launch of the halo kernel's shape that does not fill whole rounds (conv3x3_bf16.hip): descriptors and segment
// tables of the two parts.  Returns the head's image count, 0: no cut.
struct HeadTail { tg_igemm_desc dh, dt; int32_t seg_h[8], seg_t[8]; int nh, nt, k0; };
static int head_tail(const tg_igemm_desc* descs, int n_desc, const int32_t* seg_rows, int nseg, bool colsum, bool bf16, HeadTail* o) {
  const int head = tg::conv3x3_bf16_split_images(descs, n_desc, seg_rows, colsum ? nseg : 0, bf16);
  if (!head) return 0;
  o->dh = descs[0]; o->dt = descs[0];
  o->dh.n_img = head;
  o->dt.n_img = descs[0].n_img - head;
  const int64_t per_img = (int64_t)descs[0].h_in * descs[0].w_in;
  o->nh = o->nt = o->k0 = 0;
  if (colsum) {
    int64_t left = (int64_t)head * per_img;                   // rows of the head still to hand out
    for (int i = 0; i < nseg; ++i) {
      const int64_t r = seg_rows[i], take = r < left ? r : left;
      if (take > 0) o->seg_h[o->nh++] = (int32_t)take;
      if (r - take > 0) {
        if (o->nt == 0) o->k0 = i;
        o->seg_t[o->nt++] = (int32_t)(r - take);
      }
      left -= take;
    }
  }
  return head;
}

static int igemm_impl(const tg_igemm_desc* descs, int n_desc, const float* in, const float* w, const float* bias, float* out, void* stream,
                      double* colsum, const int32_t* seg_rows, int nseg, bool bf16 = false, const float* ymul = nullptr, int ymul_act = 0,
                      float ymul_alpha = 0.f, void* scratch = nullptr, int64_t scratch_bytes = 0, int stat2 = 0, const float* lab = nullptr,
                      int lab_n = 0) {
  TG_REQUIRE(descs && n_desc >= 1 && n_desc <= MAX_SUB, "igemm: n_desc=%d out of range", n_desc);
  TG_REQUIRE(in && w && out, "igemm: null buffer");
  if (lab) {
    const tg_igemm_desc* e = &descs[0];
    TG_REQUIRE(n_desc == 1 && !colsum && e->n_group == 0 && e->os_x == 1 && e->os_y == 1 && lab_n >= 1 && (e->n_store & 3) == 0 && (e->ld_out & 3) == 0 &&
               e->ld_out >= e->n_store + lab_n,
               "igemm_labels: needs one ungrouped sub-problem with os = 1, 4 | n_store, 4 | ld_out and ld_out >= n_store + n_labels (n_store %d, ld_out %d, n_labels %d)",
               e->n_store, e->ld_out, lab_n);
  }
  // A 3x3 layer of the halo kernel's shape whose launch does not fill whole rounds of one workgroup per CU (conv3x3_bf16.hip): the leading
  // images that do go to that kernel, the few left over to the generic one — two launches over disjoint image ranges of the same buffers
  // (the column sums of both accumulate into the same per-segment accumulators).
  HeadTail ht;
  if (const int head = lab ? 0 : head_tail(descs, n_desc, seg_rows, nseg, colsum != nullptr, bf16, &ht)) {
    const int64_t per_img = (int64_t)descs[0].h_in * descs[0].w_in;
    int rc = igemm_impl(&ht.dh, 1, in, w, bias, out, stream, colsum, colsum ? ht.seg_h : nullptr, ht.nh, bf16, ymul, ymul_act, ymul_alpha, scratch, scratch_bytes, stat2);
    if (rc != TG_OK) return rc;
    const int64_t o_in = (int64_t)head * per_img * descs[0].ld_in, o_out = (int64_t)head * per_img * descs[0].ld_out;
    // (the tail is stream-ordered behind the head: both may use the same scratch)
    return igemm_impl(&ht.dt, 1, in + o_in, w, bias, out + o_out, stream, colsum ? colsum + (int64_t)ht.k0 * descs[0].c_out * (stat2 ? 2 : 1) : nullptr,
                      colsum ? ht.seg_t : nullptr, ht.nt, bf16, ymul ? ymul + o_out : nullptr, ymul_act, ymul_alpha, scratch, scratch_bytes, stat2);
  }
  IgemmParams p;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.n_sub = n_desc;
  p.colsum = colsum; p.nseg = nseg;
  p.ymul = ymul; p.ymul_act = ymul_act; p.ymul_alpha = ymul_alpha;
  p.stat2 = stat2;
  p.lab = lab; p.lab_n = lab_n; p.lab_c0 = descs[0].n_store;
  for (int i = 0; i < 8; ++i) p.seg_rows[i] = (seg_rows && i < nseg) ? seg_rows[i] : 0;
  const tg_igemm_desc* d = &descs[0];
  // sub-problems longest first: workgroups are dispatched in index order, so the 9-tap parity of a 5x5 s2 transposed
  // conv starts before the 4-tap one instead of forming the tail
  int order[MAX_SUB] = {0, 1, 2, 3};
  tg::igemm_sub_order(descs, n_desc, order);
  for (int i = 0; i < n_desc; ++i) {
    int rc = check_desc(&descs[order[i]]);
    if (rc != TG_OK) return rc;
    const tg_igemm_desc& e = descs[order[i]];
    TG_REQUIRE(e.n_img == d->n_img && e.h_v == d->h_v && e.w_v == d->w_v && e.c_out == d->c_out && e.ld_in == d->ld_in && e.h_in == d->h_in &&
               e.w_in == d->w_in && e.w_sn == d->w_sn && e.w_st == d->w_st, "igemm: sub-problem %d differs in M / N / gathered tensor / filter strides", i);
    SubDesc& k = p.d[i];
    k.h_v = e.h_v; k.w_v = e.w_v; k.s_y = e.s_y; k.s_x = e.s_x; k.h_out = e.h_out; k.w_out = e.w_out; k.ld_out = e.ld_out;
    k.os_y = e.os_y; k.os_x = e.os_x; k.oo_y = e.oo_y; k.oo_x = e.oo_x; k.n_store = e.n_store; k.n_taps = e.n_taps; k.act = e.act;
    k.alpha = e.alpha;
    k.n_group = e.n_group;
    for (int t = 0; t < e.n_taps; ++t)
      k.taps[t] = ((int32_t)e.tapw[t] << 16) | (((int32_t)e.dy[t] & 0xff) << 8) | ((int32_t)e.dx[t] & 0xff);
  }
  p.w_sn = d->w_sn; p.w_st = d->w_st;
  p.n_img = d->n_img; p.h_in = d->h_in; p.w_in = d->w_in; p.ld_in = d->ld_in; p.c_out = d->c_out;
  p.M = d->n_img * d->h_v * d->w_v;
  const int64_t in_bytes = (int64_t)d->n_img * d->h_in * d->w_in * d->ld_in * 4;
  TG_REQUIRE(in_bytes < 0x7FFFFFF0LL, "igemm: gathered tensor exceeds the 2 GiB buffer-descriptor range");
  p.in_bytes = (uint32_t)in_bytes;
  int max_tapw = 0;
  for (int i = 0; i < n_desc; ++i)
    for (int t = 0; t < descs[i].n_taps; ++t) max_tapw = descs[i].tapw[t] > max_tapw ? descs[i].tapw[t] : max_tapw;
  const int64_t w_bytes = ((int64_t)(d->c_out - 1) * d->w_sn + (int64_t)max_tapw * d->w_st + d->ld_in) * 4;
  TG_REQUIRE(w_bytes > 0 && w_bytes < 0x7FFFFFF0LL && d->w_st >= 0 && d->w_sn >= 0 && (int64_t)max_tapw * d->w_st < 0x7FFFFFFFLL,
             "igemm: filter extent outside the 2 GiB buffer-descriptor range");
  p.w_bytes = (uint32_t)w_bytes;
  int64_t out_bytes = 0;
  for (int i = 0; i < n_desc; ++i) {
    const int64_t ob = (int64_t)descs[i].n_img * descs[i].h_out * descs[i].w_out * descs[i].ld_out * 4;
    out_bytes = ob > out_bytes ? ob : out_bytes;
  }
  TG_REQUIRE(out_bytes < 0x7FFFFFF0LL, "igemm: output tensor exceeds the 2 GiB buffer-descriptor range");
  p.out_bytes = (uint32_t)out_bytes;
  double taps = 0;
  for (int i = 0; i < n_desc; ++i) taps += descs[i].n_taps;
  const double flops = 2.0 * p.M * d->c_out * taps * d->ld_in;
  const double bytes = 4.0 * ((double)p.M * d->ld_in + (double)p.M * d->n_store * n_desc + (double)d->c_out * taps * d->ld_in);
  hipStream_t s = tg::as_stream(stream);
  const bool halo = !lab && tg::conv3x3_bf16_applicable(descs, n_desc, seg_rows, colsum ? nseg : 0, bf16);      // the classifier's 3x3 layers: halo-tiled kernel (both operand types)
  // tile choice by the quantisation cost model of geom.cpp (tg::igemm_pick_tile; also behind tg_igemm_tile / tg_igemm_colsum_supported)
  int bm = 0, bn = 0;
  tg::IgemmSched sc;
  std::memset(&sc, 0, sizeof sc);
  if (!halo) {
    TG_REQUIRE(tg::igemm_pick_tile(descs, n_desc, colsum != nullptr, seg_rows, nseg, bf16, &bm, &bn), "igemm: no tile fits c_out=%d with the given segments",
               d->c_out);
    // work units: tiles of the last partial round / of an under-filled launch / of the long sub-problems are cut along K when the caller
    // brought scratch for their partial sums (tg_igemm_workspace_bytes); otherwise one unit per tile
    p.m_tiles = (p.M + bm - 1) / bm;
    p.n_tiles = (p.c_out + bn - 1) / bn;
    int nk[MAX_SUB] = {0, 0, 0, 0};
    for (int i = 0; i < n_desc; ++i) nk[i] = descs[order[i]].n_taps * (d->ld_in / BK);
    static const int dbg_mask = getenv("TG_IGEMM_SPLIT_MASK") ? atoi(getenv("TG_IGEMM_SPLIT_MASK")) : 7;      // debugging aid, read once: 1 plain, 2 column-sum variants, 4 several sub-problems
    const int kind = n_desc > 1 ? 4 : (colsum ? 2 : 1);
    tg::igemm_schedule(n_desc, nk, (int64_t)p.m_tiles * p.n_tiles, bm, bn, 2 * tg::halo_compute_units(), scratch != nullptr && (dbg_mask & kind), &sc);
    if (sc.ws_bytes > scratch_bytes || (reinterpret_cast<uintptr_t>(scratch) & 15))      // less scratch than the cut needs: the one-launch schedule
      tg::igemm_schedule(n_desc, nk, (int64_t)p.m_tiles * p.n_tiles, bm, bn, 2 * tg::halo_compute_units(), false, &sc);
  }
  char desc[128];
  snprintf(desc, sizeof(desc), "M=%dx%d N=%d K=%gx%d in=%dx%d s=%d os=%d%s tile=%dx%d units=%d fix=%d ks=%d/%d/%d/%d", n_desc, p.M, d->c_out, taps, d->ld_in, d->h_in,
           d->w_in, d->s_y, d->os_y, bf16 ? " bf16" : "", bm, bn, sc.n_units, sc.n_fix, sc.ks[0], sc.ks[1], sc.ks[2], sc.ks[3]);
  tg::ProfScope prof(tg::PC_IGEMM, flops, bytes, s, desc);
  if (halo)
    return tg::conv3x3_bf16_launch(d, in, w, bias, out, colsum, seg_rows, nseg, ymul, ymul_act, ymul_alpha, p.in_bytes, p.w_bytes, p.out_bytes, s, bf16,
                                   scratch, scratch_bytes, stat2);
  p.n_units = sc.n_units; p.n_fix = sc.n_fix; p.nfull = sc.nfull; p.pat_len = sc.pat_len; p.n_pat_split = sc.n_pat_split; p.n_split_sub = sc.n_split_sub;
  for (int i = 0; i < MAX_SUB; ++i) { p.ks[i] = sc.ks[i]; p.split_sub[i] = sc.split_sub[i]; p.first_slot[i] = sc.first_slot[i]; }
  for (int i = 0; i < 16; ++i) { p.pat_sub[i] = sc.pat_sub[i]; p.pat_k[i] = sc.pat_k[i]; p.pat_slot[i] = sc.pat_slot[i]; }
  p.ws = static_cast<float*>(scratch);
  if (bm == 128 && bn == 128) launch_igemm<128, 128, 2, 2>(p, s, bf16);
  else if (bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2>(p, s, bf16);
  else if (bm == 64 && bn == 128) launch_igemm<64, 128, 2, 2>(p, s, bf16);
  else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2>(p, s, bf16);
  else if (bm == 32 && bn == 128) launch_igemm<32, 128, 1, 4>(p, s, bf16);
  else launch_igemm<128, 32, 4, 1>(p, s, bf16);
  TG_CHECK_LAUNCH("igemm_f32_kernel");
  return TG_OK;
}

#ifdef TG_STAMP
extern "C" int tg_debug_read_block_times(unsigned long long* out, int n) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_block_times), sizeof(unsigned long long) * 3 * n);
  return e == hipSuccess ? 0 : -2;
}

extern "C" int tg_debug_read_stamps(unsigned long long* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_stamps), sizeof(unsigned long long) * 64);
  return e == hipSuccess ? 0 : -2;
}
#endif

extern "C" int tg_igemm_multi_f32(const tg_igemm_desc* descs, int n_desc, const float* in, const float* w, const float* bias, float* out,
                                  void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_impl(descs, n_desc, in, w, bias, out, stream, nullptr, nullptr, 0, false, nullptr, 0, 0.f, scratch, scratch_bytes);
}

extern "C" int tg_igemm_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, void* scratch,
                            int64_t scratch_bytes, void* stream) {
  return igemm_impl(d, 1, in, w, bias, out, stream, nullptr, nullptr, 0, false, nullptr, 0, 0.f, scratch, scratch_bytes);
}

extern "C" int tg_igemm_labels_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, const float* labels, int n_labels, float* out,
                                   void* scratch, int64_t scratch_bytes, void* stream) {
  TG_REQUIRE(labels != nullptr, "igemm_labels: null labels");
  return igemm_impl(d, 1, in, w, bias, out, stream, nullptr, nullptr, 0, false, nullptr, 0, 0.f, scratch, scratch_bytes, 0, labels, n_labels);
}

extern "C" int tg_igemm_labels_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, const float* labels, int n_labels, float* out,
                                    void* scratch, int64_t scratch_bytes, void* stream) {
  TG_REQUIRE(labels != nullptr, "igemm_labels: null labels");
  return igemm_impl(d, 1, in, w, bias, out, stream, nullptr, nullptr, 0, true, nullptr, 0, 0.f, scratch, scratch_bytes, 0, labels, n_labels);
}

// scratch a launch of these descriptors can use: the larger of the halo kernel's packed bf16 filter (REQUIRED by such a launch) and the
// partial sums of the tiles the generic kernel's schedule cuts along K (optional: with less the launch runs as one unit per tile);
// a launch that is cut into a halo head and a generic tail uses the scratch for one after the other.  Mirrors the routing of igemm_impl.
static int64_t igemm_ws_bytes(const tg_igemm_desc* descs, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  const bool colsum = nseg > 0;
  HeadTail ht;
  if (head_tail(descs, n_desc, seg_rows, nseg, colsum, bf16, &ht)) {
    const int64_t a = igemm_ws_bytes(&ht.dh, 1, colsum ? ht.seg_h : nullptr, ht.nh, bf16);
    const int64_t b = igemm_ws_bytes(&ht.dt, 1, colsum ? ht.seg_t : nullptr, ht.nt, bf16);
    return a > b ? a : b;
  }
  if (tg::conv3x3_bf16_applicable(descs, n_desc, seg_rows, colsum ? nseg : 0, bf16)) return bf16 ? tg::conv3x3_bf16_pack_bytes(descs, n_desc) : 0;
  int bm = 0, bn = 0;
  if (!tg::igemm_pick_tile(descs, n_desc, colsum, seg_rows, nseg, bf16, &bm, &bn)) return 0;
  int order[MAX_SUB] = {0, 1, 2, 3}, nk[MAX_SUB] = {0, 0, 0, 0};
  tg::igemm_sub_order(descs, n_desc, order);
  for (int i = 0; i < n_desc; ++i) nk[i] = descs[order[i]].n_taps * (descs[0].ld_in / BK);
  const int64_t M = (int64_t)descs[0].n_img * descs[0].h_v * descs[0].w_v;
  tg::IgemmSched sc;
  tg::igemm_schedule(n_desc, nk, ((M + bm - 1) / bm) * ((descs[0].c_out + bn - 1) / bn), bm, bn, 2 * tg::halo_compute_units(), true, &sc);
  return sc.ws_bytes;
}

extern "C" int64_t tg_igemm_workspace_bytes(const tg_igemm_desc* descs, int n_desc, const int32_t* seg_rows, int nseg, int bf16) {
  if (!descs || n_desc < 1 || n_desc > MAX_SUB || nseg < 0 || nseg > 8 || (nseg > 0 && !seg_rows)) { tg::set_error("igemm_workspace_bytes: bad arguments"); return TG_ERR_INVALID; }
  for (int i = 0; i < n_desc; ++i)
    if (descs[i].ld_in <= 0 || descs[i].c_out <= 0 || descs[i].n_taps <= 0 || descs[i].n_taps > TG_MAX_TAPS || descs[i].n_img <= 0) { tg::set_error("igemm_workspace_bytes: bad descriptor"); return TG_ERR_INVALID; }
  return igemm_ws_bytes(descs, n_desc, seg_rows, nseg, bf16 != 0);
}

static int igemm_colsum_impl(const tg_igemm_desc* d, const float* in, const float* w, float* out, const int32_t* seg_rows, int nseg,
                             double* colsum, int colsum_zeroed, void* stream, bool bf16, const float* ymul = nullptr, int ymul_act = 0,
                             float ymul_alpha = 0.f, void* scratch = nullptr, int64_t scratch_bytes = 0) {
  TG_REQUIRE(d && colsum && seg_rows && nseg >= 1 && nseg <= 8, "igemm_colsum: bad args");
  TG_REQUIRE(d->n_group == 0, "igemm_colsum: grouped columns are not supported");
  TG_REQUIRE(d->act == TG_ACT_NONE, "igemm_colsum: the statistics are of the raw convolution output (no activation)");
  int tot = 0;
  for (int i = 0; i < nseg; ++i) { TG_REQUIRE(seg_rows[i] >= 32, "igemm_colsum: segment %d has %d rows (need at least one 32-row tile)", i, seg_rows[i]); tot += seg_rows[i]; }
  TG_REQUIRE(tot == d->n_img * d->h_v * d->w_v, "igemm_colsum: segments sum to %d rows, launch has %d", tot, d->n_img * d->h_v * d->w_v);
  if (!colsum_zeroed) {
    hipError_t e = hipMemsetAsync(colsum, 0, sizeof(double) * nseg * d->c_out, tg::as_stream(stream));
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(colsum)");
  }
  return igemm_impl(d, 1, in, w, nullptr, out, stream, colsum, seg_rows, nseg, bf16, ymul, ymul_act, ymul_alpha, scratch, scratch_bytes);
}

extern "C" int tg_igemm_colsum_f32(const tg_igemm_desc* d, const float* in, const float* w, float* out, const int32_t* seg_rows, int nseg,
                                   double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_colsum_impl(d, in, w, out, seg_rows, nseg, colsum, colsum_zeroed, stream, false, nullptr, 0, 0.f, scratch, scratch_bytes);
}

extern "C" int tg_igemm_colsum_bf16(const tg_igemm_desc* d, const float* in, const float* w, float* out, const int32_t* seg_rows, int nseg,
                                    double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_colsum_impl(d, in, w, out, seg_rows, nseg, colsum, colsum_zeroed, stream, true, nullptr, 0, 0.f, scratch, scratch_bytes);
}

// conv + bias + activation whose output feeds a batch norm: the statistics of the ACTIVATED output are taken in the epilogue
static int igemm_bnstat_impl(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, const int32_t* seg_rows, int nseg,
                             double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream, bool bf16) {
  TG_REQUIRE(d && sums && seg_rows && nseg >= 1 && nseg <= 8, "igemm_bnstat: bad args");
  TG_REQUIRE(d->n_group == 0, "igemm_bnstat: grouped columns are not supported");
  TG_REQUIRE(d->act == TG_ACT_NONE || d->act == TG_ACT_RELU || d->act == TG_ACT_LRELU, "igemm_bnstat: activation %d is not none / relu / leaky relu", d->act);
  int tot = 0;
  for (int i = 0; i < nseg; ++i) { TG_REQUIRE(seg_rows[i] >= 32, "igemm_bnstat: segment %d has %d rows (need at least one 32-row tile)", i, seg_rows[i]); tot += seg_rows[i]; }
  TG_REQUIRE(tot == d->n_img * d->h_v * d->w_v, "igemm_bnstat: segments sum to %d rows, launch has %d", tot, d->n_img * d->h_v * d->w_v);
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 8 * 2 * nseg * d->c_out, tg::as_stream(stream));      // all eight replicas of the batch norm's buffer
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(bn statistics)");
  }
  return igemm_impl(d, 1, in, w, bias, out, stream, sums, seg_rows, nseg, bf16, nullptr, 0, 0.f, scratch, scratch_bytes, 1);
}

extern "C" int tg_igemm_bnstat_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, const int32_t* seg_rows, int nseg,
                                   double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_bnstat_impl(d, in, w, bias, out, seg_rows, nseg, sums, sums_zeroed, scratch, scratch_bytes, stream, false);
}

extern "C" int tg_igemm_bnstat_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, const int32_t* seg_rows, int nseg,
                                    double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_bnstat_impl(d, in, w, bias, out, seg_rows, nseg, sums, sums_zeroed, scratch, scratch_bytes, stream, true);
}

// input gradient (or any conv) whose output dy is consumed by a batch norm's backward pass: that pass's statistics in the epilogue
static int igemm_bnbwdstat_impl(const tg_igemm_desc* d, const float* in, const float* w, const float* x, float* out, const int32_t* seg_rows, int nseg,
                                double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream, bool bf16) {
  TG_REQUIRE(d && sums && x && seg_rows && nseg >= 1 && nseg <= 8, "igemm_bnbwdstat: bad args");
  TG_REQUIRE(d->n_group == 0 && d->act == TG_ACT_NONE, "igemm_bnbwdstat: grouped columns / a fused activation are not supported");
  int tot = 0;
  for (int i = 0; i < nseg; ++i) { TG_REQUIRE(seg_rows[i] >= 32, "igemm_bnbwdstat: segment %d has %d rows (need at least one 32-row tile)", i, seg_rows[i]); tot += seg_rows[i]; }
  TG_REQUIRE(tot == d->n_img * d->h_v * d->w_v, "igemm_bnbwdstat: segments sum to %d rows, launch has %d", tot, d->n_img * d->h_v * d->w_v);
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 8 * 2 * nseg * d->c_out, tg::as_stream(stream));      // all eight replicas of the batch norm's buffer
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(bn backward statistics)");
  }
  return igemm_impl(d, 1, in, w, nullptr, out, stream, sums, seg_rows, nseg, bf16, x, TG_ACT_NONE, 0.f, scratch, scratch_bytes, 2);
}

extern "C" int tg_igemm_bnbwdstat_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* x, float* out, const int32_t* seg_rows, int nseg,
                                      double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_bnbwdstat_impl(d, in, w, x, out, seg_rows, nseg, sums, sums_zeroed, scratch, scratch_bytes, stream, false);
}

extern "C" int tg_igemm_bnbwdstat_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* x, float* out, const int32_t* seg_rows, int nseg,
                                       double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_bnbwdstat_impl(d, in, w, x, out, seg_rows, nseg, sums, sums_zeroed, scratch, scratch_bytes, stream, true);
}

extern "C" int tg_igemm_actsum_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* yact, int act, float alpha, float* out,
                                   const int32_t* seg_rows, int nseg, double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes,
                                   void* stream) {
  TG_REQUIRE(yact != nullptr, "igemm_actsum: yact is NULL");
  return igemm_colsum_impl(d, in, w, out, seg_rows, nseg, colsum, colsum_zeroed, stream, false, yact, act, alpha, scratch, scratch_bytes);
}

extern "C" int tg_igemm_actsum_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* yact, int act, float alpha, float* out,
                                    const int32_t* seg_rows, int nseg, double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes,
                                    void* stream) {
  TG_REQUIRE(yact != nullptr, "igemm_actsum: yact is NULL");
  return igemm_colsum_impl(d, in, w, out, seg_rows, nseg, colsum, colsum_zeroed, stream, true, yact, act, alpha, scratch, scratch_bytes);
}

extern "C" int tg_igemm_multi_bf16(const tg_igemm_desc* descs, int n_desc, const float* in, const float* w, const float* bias, float* out,
                                   void* scratch, int64_t scratch_bytes, void* stream) {
  return igemm_impl(descs, n_desc, in, w, bias, out, stream, nullptr, nullptr, 0, true, nullptr, 0, 0.f, scratch, scratch_bytes);
}

extern "C" int tg_igemm_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, void* scratch,
                             int64_t scratch_bytes, void* stream) {
  return igemm_impl(d, 1, in, w, bias, out, stream, nullptr, nullptr, 0, true, nullptr, 0, 0.f, scratch, scratch_bytes);
}

template <int CT, int NT, int WC, int WN, int WK>
static void launch_wgrad(WgradParams& p, hipStream_t s, bool bf16) {
  p.c_tiles = (p.d.ld_in + CT - 1) / CT;      // last tiles may overhang (masked at the store)
  p.n_tiles = (p.d.c_out + NT - 1) / NT;
  int blocks = p.n_split * p.d.n_taps * p.c_tiles * p.n_tiles;
  if (bf16) hipLaunchKernelGGL((wgrad_f32_kernel<CT, NT, WC, WN, WK, true>), dim3(blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((wgrad_f32_kernel<CT, NT, WC, WN, WK, false>), dim3(blocks), dim3(256), 0, s, p);
}

static int wgrad_impl(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream, bool bf16) {
  int rc = check_desc(d);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(in && dout && slab, "wgrad: null buffer");
  TG_REQUIRE(n_split >= 1, "wgrad: n_split=%d", n_split);
  TG_REQUIRE(d->c_out <= d->ld_out, "wgrad: c_out=%d exceeds ld_out=%d", d->c_out, d->ld_out);
  TG_REQUIRE(d->n_group == 0, "wgrad: grouped columns are not supported");
  WgradParams p{in, dout, slab, *d, 0, n_split, 0, 0, 0, 0, 0};
  p.M = d->n_img * d->h_v * d->w_v;
  const int64_t ib = (int64_t)d->n_img * d->h_in * d->w_in * d->ld_in * 4, ob = (int64_t)d->n_img * d->h_out * d->w_out * d->ld_out * 4;
  TG_REQUIRE(ib < 0x7FFFFFF0LL && ob < 0x7FFFFFF0LL, "wgrad: operand exceeds the 2 GiB buffer-descriptor range");
  p.in_bytes = (uint32_t)ib; p.dout_bytes = (uint32_t)ob;
  p.px_per_split = (((p.M + n_split - 1) / n_split) + BK - 1) / BK * BK;
  const double flops = 2.0 * p.M * d->c_out * d->n_taps * d->ld_in;
  const double bytes = 4.0 * ((double)p.M * d->ld_in + (double)p.M * d->c_out + (double)n_split * d->c_out * d->n_taps * d->ld_in);
  char desc[96];
  snprintf(desc, sizeof(desc), "M=%d N=%d K=%dx%d in=%dx%d s=%d split=%d%s", p.M, d->c_out, d->n_taps, d->ld_in, d->h_in, d->w_in, d->s_y, n_split,
           bf16 ? " bf16" : "");
  tg::ProfScope prof(tg::PC_WGRAD, flops, bytes, tg::as_stream(stream), desc);
  hipStream_t s = tg::as_stream(stream);
  if (tg::wgrad3x3_applicable(d, n_split, bf16, tg::halo_policy(), tg::halo_compute_units())) {      // the classifier's 3x3 layers: wgrad3x3.hip
    tg::halo_count_launch();
    return tg::wgrad3x3_launch(d, in, dout, slab, n_split, p.in_bytes, p.dout_bytes, s, bf16);
  }
  // channel tiles: tg::wgrad_tile (geom.cpp; tg_wgrad_splits sizes the pixel split with the same rule)
  const int ct = tg::wgrad_tile(d->ld_in), nt = tg::wgrad_tile(d->c_out);
  if (ct == 128 && nt == 128) launch_wgrad<128, 128, 2, 2, 1>(p, s, bf16);
  else if (ct == 128 && nt == 64) launch_wgrad<128, 64, 2, 2, 1>(p, s, bf16);
  else if (ct == 128 && nt == 32) launch_wgrad<128, 32, 4, 1, 1>(p, s, bf16);
  else if (ct == 64 && nt == 128) launch_wgrad<64, 128, 2, 2, 1>(p, s, bf16);
  else if (ct == 64 && nt == 64) launch_wgrad<64, 64, 2, 2, 1>(p, s, bf16);
  else if (ct == 64 && nt == 32) launch_wgrad<64, 32, 2, 1, 2>(p, s, bf16);
  else if (ct == 32 && nt == 128) launch_wgrad<32, 128, 1, 4, 1>(p, s, bf16);
  else if (ct == 32 && nt == 64) launch_wgrad<32, 64, 1, 2, 2>(p, s, bf16);
  else launch_wgrad<32, 32, 1, 1, 4>(p, s, bf16);
  TG_CHECK_LAUNCH("wgrad_f32_kernel");
  return TG_OK;
}

extern "C" int tg_wgrad_f32(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream) {
  return wgrad_impl(d, in, dout, slab, n_split, stream, false);
}

extern "C" int tg_wgrad_bf16(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream) {
  return wgrad_impl(d, in, dout, slab, n_split, stream, true);
}
