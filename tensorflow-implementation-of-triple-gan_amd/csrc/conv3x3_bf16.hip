// 3x3 / stride 1 / SAME convolution on the bf16 matrix cores of gfx950 (BASELINE.json configs[3], "bf16 MFMA conv path"):
// the fast path behind tg_igemm_bf16 / tg_igemm_colsum_bf16 / tg_igemm_actsum_bf16 for the classifier's 3x3 layers — forward AND
// input gradient (the same geometry with the transposed filter layout; the weight strides come from the descriptor).
//
// Why not the generic implicit GEMM (igemm.hip) with bf16 operands: gathering one K-tile per (tap, channel chunk) re-reads every input
// pixel nine times through L1/L2, and at the bf16 MFMA rate (1024 FLOP/clk/SIMD, 16x the fp32 one) that path — not the matrix pipe —
// bounds the kernel (round 1: 200 TFLOP/s = 8 % of peak with fp32 LDS images converted fragment by fragment).  Here
//   * a workgroup owns 256 output pixels = R whole image rows (R*W = 256, W in {16, 32, 64}) x 128 output channels;
//   * per 64-channel chunk the (R+2) x (W+2) input HALO is loaded ONCE (fp32 in HBM -> v_cvt_pk_bf16_f32 -> bf16 in LDS; pixels outside
//     the image are out-of-range buffer loads, i.e. hardware zeros) and all nine taps read their fragments from it at shifted pixel
//     addresses: operand traffic from L2 drops ~7x, the LDS image is half the size, one ds_read_b128 feeds a whole 32x32x16 fragment;
//   * the filter tile of each (tap, chunk) — 128 rows x 64 channels — is converted on its global -> LDS path the same way;
//   * LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with bits 1..3 of the row so that every 16-lane group of a
//     ds_read_b128 (and every ds_write_b64 of the staging pass) hits 16 different bank slots;
//   * 8 waves (4 along pixels x 2 along channels, 64x64 outputs each = 2x2 v_mfma_f32_32x32x16_bf16 tiles), two per SIMD; a K-step is a
//     group of THREE taps of one chunk (48 MFMAs per wave = 3 072 matrix-pipe cycles per SIMD): the filter loads of the next group are
//     issued before them and written to the other LDS buffer after them, so their L2 latency has a whole group to hide behind; one
//     barrier per group (with one tap per barrier the 16 MFMAs of a wave were too short for that: 3.5 k cycles per 1 k of matrix work).
// Numerics: every MFMA operand is rounded to bf16 (RNE) exactly as the generic tg_*_bf16 kernels do, products accumulate in fp32 —
// the results differ from those kernels only by the order of the fp32 accumulation (channel chunks outermost here).
#include <cstdlib>
#include "tg_common.h"
#include "tg_device.h"
#include "tg_conv3x3_bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef TG_STAMP
// Diagnostic build (csrc/Makefile target libtg_stamp.so, never shipped): cycle stamps of workgroup phases, wave 0 of a few workgroups,
// read back with tg_debug_read_conv_stamps.  Stamp values go only to this buffer; no output depends on them.
__device__ unsigned long long tg_conv_stamps[8 * 64];
#define CSTAMP(i)                                                                                        \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 64) {                                                           \
      unsigned long long t_;                                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
      tg_conv_stamps[(i) * 64 + blockIdx.x] = t_;                                                        \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
#define LSTAMP(var)                                                                \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");    \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define CSTAMP(i) do {} while (0)
#define LSTAMP(var) do {} while (0)
#endif

#ifndef TG_ABL
#define TG_ABL 0                               // timing ablations of conv3x3_pipe_kernel (tools only, results then wrong): 1 no stores, 8 half the halo loads, 16 half the filter loads, 32 no MFMAs
#endif

namespace {

constexpr int BN = 128, KC = 64;               // output channels per tile, channels per chunk
constexpr uint32_t OOB = 0x80000000u;          // byte offset beyond any (< 2 GiB) tensor: buffer loads return 0, stores are dropped

struct ConvParams {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  double* colsum;
  const float* ymul;
  int ymul_act;
  float ymul_alpha;
  int nseg, seg_rows[8];
  int n_img, h, ld_in, ld_out, c_out, n_store, act;
  float alpha;
  int64_t w_sn, w_st;                          // filter element (n, tap, c) at n*w_sn + tapw[tap]*w_st + c
  int tap[9];                                  // (tapw << 16) | ((dy & 0xff) << 8) | (dx & 0xff)
  int n_tiles_m, n_tiles_n;
  uint32_t in_bytes, w_bytes, out_bytes;
  const void* wpk;                             // bf16 filter packed as consecutive LDS images (filter_pack_kernel) — conv3x3_pipe_kernel<.., BF16 = true>
  int f32;                                     // 1: exact-fp32 operands (conv3x3_ws_kernel<..., BF16 = false>)
  int stagger;                                 // first-round delay (in units of ~1k cycles) of every second workgroup slot: see the kernel
  int dbg;                                     // TG_CONV3X3_DBG (diagnostic timing only, results then wrong): 1 no stores, 2 no filter loads in the loop, 4 no MFMAs, 8 no halo reload
};

int compute_units() {
  static const int n = ([] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; })();
  return n;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) {   // 2 fp32 -> 2 bf16 (RNE): one v_cvt_pk_bf16_f32
  const f32x2 f = {__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

__device__ __forceinline__ u32x2 pack4(u32x4 v) {             // 4 fp32 -> 4 bf16 (RNE), 8 bytes, channel order kept
  u32x2 r;
  r.x = pack2(v.x, v.y);
  r.y = pack2(v.z, v.w);
  return r;
}

// v + (v of the lane the DPP control names); lanes the row mask excludes add 0
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// LDS byte offset of 16-B chunk `chunk` (8 bf16) of row `row`: 128-B rows, chunk index XOR-swizzled with bits 1..3 of the row.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// BM pixels per tile (256: 8 waves, 128: 4 waves — 64 pixels x 64 channels per wave either way), TPS taps per barrier step (3 or 1).
template <int W, int BM, int TPS, bool COLSUM>
__global__ void __launch_bounds__(2 * BM, 2) conv3x3_bf16_kernel(ConvParams p) {
  constexpr int THREADS = 2 * BM;
  constexpr int R = BM / W;                       // image rows per tile
  constexpr int HW_ = W + 2, HP = (R + 2) * HW_;  // halo geometry
  constexpr int A_BYTES = (HP * 128 + 255) / 256 * 256, B_TAP = BN * 128, B_BYTES = TPS * B_TAP;   // one halo chunk; the filter tiles of TPS taps
  constexpr int A_UNITS = HP * 16, A_IT = (A_UNITS + THREADS - 1) / THREADS;   // 16-B fp32 loads of one halo chunk, per thread
  constexpr int B_IT = BN * 16 / THREADS;                                      // = 4 per tap
  constexpr int EPI_BYTES = 128 * (BN + 4) * 4 + (THREADS / BN) * BN * 4;
  constexpr int MAIN_BYTES = A_BYTES + 2 * B_BYTES;
  constexpr int SMEM = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + BM * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;
  uint32_t* t_out = reinterpret_cast<uint32_t*>(smem + (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Stagger: all workgroups of a launch start together and would meet in their load / store phases (HBM-bound, matrix pipes idle) and
  // in their K loops (matrix-bound, HBM idle) at the same time.  Half of the first-round workgroups therefore start late by about
  // half a tile time, so that from then on the two halves alternate between the memory and the matrix phases.
  if (p.stagger > 0) {
    const int slot = blockIdx.x >> 3;                                  // index inside its XCD (blocks are dealt round-robin over 8 XCDs)
    const bool late = (2 * BM == 512) ? (slot & 1) && blockIdx.x < 256 : (slot >= 32 && slot < 64);
    if (late)
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);   // ~1k cycles each
  }
  CSTAMP(0);
  const int lid = xcd_remap(blockIdx.x, p.n_tiles_m * p.n_tiles_n);
  const int nt = lid % p.n_tiles_n, mt = lid / p.n_tiles_n;
  const int n0 = nt * BN;
  const int tiles_per_img = p.h / R;
  const int img = mt / tiles_per_img, row0 = (mt - img * tiles_per_img) * R;     // first image row of the tile

  if (tid < BM) {
    const int ty = tid / W, tx = tid - ty * W;
    t_out[tid] = (uint32_t)(((img * p.h + row0 + ty) * W + tx) * p.ld_out) * 4u;
  }

  // ---- staging addresses.  Unit u = tid + 512*i of a staging pass is the 16-B (4 x fp32) piece q = u & 15 of row u >> 4; the row's
  // swizzle term depends on (row >> 1) & 7 only and rows advance by 32 per pass, so the LDS address of pass i is that of pass 0 plus
  // i * 32 rows; the channel chunk / tap / pass rides in the SCALAR offset of the buffer loads wherever it is uniform. ---------------
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  const int q16 = tid & 15, row_t = tid >> 4;                 // this thread's piece and first row of every staging pass
  const int st_lds = lds_off(row_t, q16 >> 1) + (q16 & 1) * 8;
  uint32_t a_voff[A_IT];                                      // halo pixels map to image pixels (or to "outside": hardware zeros)
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int hp = row_t + (THREADS / 16) * i;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int iy = row0 + hy - 1, ix = hx - 1;
    const bool ok = hp < HP && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)W;
    a_voff[i] = ok ? (uint32_t)(((img * p.h + iy) * W + ix) * p.ld_in + 4 * q16) * 4u : OOB;
  }
  const uint32_t b_voff = (uint32_t)(((int64_t)(n0 + row_t) * p.w_sn + 4 * q16) * 4);
  const uint32_t b_pass = (uint32_t)((THREADS / 16) * p.w_sn * 4);        // filter rows advance by THREADS / 16 per pass

  u32x4 ra[A_IT], rb[TPS * B_IT];
  auto gload_a = [&](int c0) {
    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_voff[i], so, 0);
  };
  auto sstore_a = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (i + 1 < A_IT || row_t + (THREADS / 16) * i < HP) *reinterpret_cast<u32x2*>(As + st_lds + i * (THREADS / 16) * 128) = pack4(ra[i]);
  };
  auto gload_b = [&](int g, int c0) {                         // the filter tiles of taps TPS*g ... TPS*g + TPS-1 for channel chunk c0
#pragma unroll
    for (int k = 0; k < TPS; ++k) {
      const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((p.tap[TPS * g + k] >> 16) * (int)p.w_st + c0) * 4);
#pragma unroll
      for (int j = 0; j < B_IT; ++j) rb[k * B_IT + j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff, so + j * b_pass, 0);
    }
  };
  auto sstore_b = [&](int buf) {
    unsigned char* b = Bs + buf * B_BYTES + st_lds;
#pragma unroll
    for (int k = 0; k < TPS; ++k)
#pragma unroll
      for (int j = 0; j < B_IT; ++j) *reinterpret_cast<u32x2*>(b + k * B_TAP + j * (THREADS / 16) * 128) = pack4(rb[k * B_IT + j]);
  };

  // ---- fragment addresses --------------------------------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;                  // (BM / 64) x 2 waves; each owns 64 pixels x 64 channels
  const int wm0 = wm * 64, wn0 = wn * 64;
  const int half = lane >> 5, col = lane & 31;
  int a_hp[2];                                              // halo pixel of (tile row, tap (0,0)) per 32-row sub-tile
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm0 + mi * 32 + col;
    const int ty = r / W, tx = r - ty * W;
    a_hp[mi] = (ty + 1) * HW_ + tx + 1;
  }
  int b_off[2][4];                                          // filter rows are fixed per wave: all four k16-step addresses up front
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int s = 0; s < 4; ++s) b_off[ni][s] = lds_off(wn0 + ni * 32 + col, 2 * s + half);

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nchunks = p.ld_in / KC;
  // prologue: halo of chunk 0, filter tiles of (taps 0..2, chunk 0)
  gload_a(0);
  gload_b(0, 0);
  CSTAMP(1);
  sstore_a();
  sstore_b(0);
  __syncthreads();
  CSTAMP(2);

  // A K-step is one GROUP of three taps of one channel chunk: 48 MFMAs per wave (1 536 matrix-pipe cycles, 3 072 per SIMD with its
  // two waves) behind which the 12 filter loads of the next group have time to arrive; they are issued first and written to the other
  // filter buffer after the MFMAs.  The next halo is issued in the second group of a chunk and written — into the ONE halo buffer —
  // behind a barrier of its own after the third.
  int bbuf = 0;
#ifdef TG_STAMP
  unsigned long long l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, a_load = 0, a_mfma = 0, a_store = 0, a_bar = 0;
#endif
  for (int c = 0; c < nchunks; ++c) {
    const bool more_c = c + 1 < nchunks;
    constexpr int NG = 9 / TPS;                             // barrier steps per chunk
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      const bool more = g < NG - 1 || more_c;
      LSTAMP(l0);
      if (more && !(p.dbg & 2)) gload_b(g < NG - 1 ? g + 1 : 0, g < NG - 1 ? c * KC : (c + 1) * KC);
      if (g == NG / 2 && more_c && !(p.dbg & 8)) gload_a((c + 1) * KC);
      LSTAMP(l1);
      const unsigned char* B = Bs + bbuf * B_BYTES;
#pragma unroll
      for (int k = 0; k < TPS; ++k) {
        const int tp = p.tap[TPS * g + k];
        const int shift = (int)(int8_t)(tp >> 8) * HW_ + (int)(int8_t)tp;
        int a_row[2], a_swz[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int hp = a_hp[mi] + shift;
          a_row[mi] = hp * 128;
          a_swz[mi] = (hp >> 1) & 7;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bf16x8 a[2], b[2];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(B + k * B_TAP + b_off[ni][s]);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              if (!(p.dbg & 4)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // D[pixel][channel]: lane = channel
              else asm volatile("" :: "v"(a[mi]), "v"(b[ni]));
        }
      }
      LSTAMP(l2);
      if (more && !(p.dbg & 2)) sstore_b(bbuf ^ 1);
      LSTAMP(l3);
      __syncthreads();
      LSTAMP(l4);
#ifdef TG_STAMP
      a_load += l1 - l0; a_mfma += l2 - l1; a_store += l3 - l2; a_bar += l4 - l3;
#endif
      if (g == NG - 1 && more_c && !(p.dbg & 8)) {                            // every wave is done with this chunk's halo: overwrite it
        sstore_a();
        __syncthreads();
      }
      bbuf ^= 1;
    }
  }

  // ---- epilogue (the operand tiles are dead; t_out lies behind them) ---------------------------------------------------------------
#ifdef TG_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 64) {
    tg_conv_stamps[5 * 64 + blockIdx.x] = a_load; tg_conv_stamps[6 * 64 + blockIdx.x] = a_mfma;
    tg_conv_stamps[7 * 64 + blockIdx.x] = (a_store << 32) | (a_bar & 0xffffffffull);
  }
#endif
  CSTAMP(3);
  // The tile leaves through LDS row-wise in two passes of 128 rows (67 KB each): a lane of the accumulator holds ONE channel and 16
  // pixels, so storing from registers would scatter dwords; from LDS every thread moves 16-B pieces of pixel rows — coalesced stores,
  // coalesced loads of the activation for the actsum form — with bias + activation applied on the way (plain form) or, for the
  // mean-only-BN forms (tg_igemm_colsum_bf16 / tg_igemm_actsum_bf16), the per-application column sums taken from LDS as well.  A tile
  // lies inside ONE image, hence inside one application segment.
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  constexpr int TLD = BN + 4, PARTS = THREADS / BN;
  float* tile = reinterpret_cast<float*>(smem);
  float* red = tile + 128 * TLD;
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ymul ? p.ymul : p.out), 0, p.out_bytes, 0x00020000);
  int seg = 0;
  if (COLSUM) {
    const int m0 = mt * BM;
    int acc_rows = p.seg_rows[0];
    while (seg < p.nseg - 1 && m0 >= acc_rows) acc_rows += p.seg_rows[++seg];
  }
  const bool ym = COLSUM && p.ymul != nullptr;
  float csum = 0.f;
  for (int pass = 0; pass < BM / 128; ++pass) {
    if ((wm >> 1) == pass) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tile[((wm & 1) * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TLD + wn0 + ni * 32 + col] = acc[mi][ni][r];
    }
    __syncthreads();
    constexpr int G4 = BN / 4;
    for (int i = tid; i < 128 * G4; i += THREADS) {
      const int rl = i / G4, cg = i - rl * G4;
      const int n = n0 + cg * 4;
      const uint32_t off = (n >= p.n_store || (p.dbg & 1)) ? OOB : t_out[pass * 128 + rl] + (uint32_t)n * 4u;
      const float4 tv = *reinterpret_cast<const float4*>(tile + rl * TLD + cg * 4);
      float va[4] = {tv.x, tv.y, tv.z, tv.w};
      if (!COLSUM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (p.bias != nullptr && n + e < p.n_store) va[e] += p.bias[n + e];
          va[e] = tgd::act(va[e], p.act, p.alpha);
        }
      } else if (ym) {      // input gradient times the activation derivative of the layer that produced this conv's input
        const u32x4 yb = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
        // (elements copied to scalars first: __builtin_bit_cast applied to a vector-element expression reads element 0 — hipcc, ROCm 7.2)
        const uint32_t y0 = yb.x, y1 = yb.y, y2 = yb.z, y3 = yb.w;
        va[0] *= tgd::act_grad(__builtin_bit_cast(float, y0), p.ymul_act, p.ymul_alpha);
        va[1] *= tgd::act_grad(__builtin_bit_cast(float, y1), p.ymul_act, p.ymul_alpha);
        va[2] *= tgd::act_grad(__builtin_bit_cast(float, y2), p.ymul_act, p.ymul_alpha);
        va[3] *= tgd::act_grad(__builtin_bit_cast(float, y3), p.ymul_act, p.ymul_alpha);
        *reinterpret_cast<float4*>(tile + rl * TLD + cg * 4) = make_float4(va[0], va[1], va[2], va[3]);
      }
      if ((p.n_store & 3) == 0) {
        const u32x4 pk = {__builtin_bit_cast(uint32_t, va[0]), __builtin_bit_cast(uint32_t, va[1]), __builtin_bit_cast(uint32_t, va[2]),
                          __builtin_bit_cast(uint32_t, va[3])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_o, off, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, va[e]), rs_o, n + e >= p.n_store ? OOB : off + 4u * e, 0, 0);
      }
    }
    if (COLSUM) {
      if (ym) __syncthreads();
      {
        const int c = tid % BN, q = tid / BN;
        float s1 = 0.f;
        for (int rl = q; rl < 128; rl += PARTS) s1 += tile[rl * TLD + c];
        red[q * BN + c] = s1;
      }
      __syncthreads();
      if (tid < BN) {
#pragma unroll
        for (int q = 0; q < PARTS; ++q) csum += red[q * BN + tid];
      }
    }
    __syncthreads();                                          // the tile is rewritten by the second pass
  }
  CSTAMP(4);
  if (COLSUM && tid < BN && n0 + tid < p.n_store) atomicAdd(p.colsum + (int64_t)seg * p.c_out + n0 + tid, (double)csum);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Role-specialised form of the 256-pixel tile (the default): waves 0-3 are CONSUMERS — one per SIMD, 64 pixels x all 128 channels each
// (2 x 4 MFMA tiles: 96 MFMAs = 3 072 matrix-pipe cycles per three-tap step, six ds_read_b128 per eight MFMAs) — and never touch
// global memory inside the K loop; waves 4-7 are LOADERS — the partner wave on each SIMD — that fetch, convert and write the next filter
// group and the next halo while the consumers multiply.  In the symmetric form above both waves of a SIMD run the same phases in
// lockstep, so the matrix pipe idles whenever they issue loads, convert, write LDS or wait at the barrier (stamped: 3.0 k of every
// 7.0 k cycles); here the pipe's wave only ever waits at the one barrier per step.  Both roles execute the same barriers.
// ---------------------------------------------------------------------------------------------------------------------------------
// BF16 = false: the SAME structure on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32) for the fp32 training step — fp32 LDS
// images of 32 channels per chunk (the same 128-B rows and swizzle), one ds_read_b128 feeding four k-steps of both operands (lane half h
// of k-group g supplies k = 8g + 4h + s in step s), no conversion.  At 1/16 of the bf16 matrix rate a three-tap step is 24.6 k matrix-pipe
// cycles per consumer, so the loaders' work and the prologue / epilogue weigh a sixteenth of what they do above.
template <int W, bool COLSUM, bool BF16>
__global__ void __launch_bounds__(512, 2) conv3x3_ws_kernel(ConvParams p) {
  constexpr int BM = 256, THREADS = 512, TPS = 3, NG = 3;
  constexpr int KCH = BF16 ? 64 : 32;                                  // channels per chunk: 128 B of LDS per pixel either way
  constexpr int UPR = BF16 ? 16 : 8;                                   // 16-B global loads (4 fp32) per row and chunk
  constexpr int RPP = 256 / UPR;                                       // rows per staging pass of the 256 loader threads
  constexpr int R = BM / W, HW_ = W + 2, HP = (R + 2) * HW_;
  constexpr int A_BYTES = (HP * 128 + 255) / 256 * 256, B_TAP = BN * 128, B_BYTES = TPS * B_TAP;
  constexpr int A_IT = (HP + RPP - 1) / RPP, B_IT = BN / RPP;
  constexpr int EPI_BYTES = 128 * (BN + 4) * 4 + (THREADS / BN) * BN * 4;
  constexpr int MAIN_BYTES = A_BYTES + 2 * B_BYTES;
  constexpr int SMEM = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + BM * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;
  uint32_t* t_out = reinterpret_cast<uint32_t*>(smem + (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  CSTAMP(0);
  const int lid = xcd_remap(blockIdx.x, p.n_tiles_m * p.n_tiles_n);
  const int nt = lid % p.n_tiles_n, mt = lid / p.n_tiles_n;
  const int n0 = nt * BN;
  const int tiles_per_img = p.h / R;
  const int img = mt / tiles_per_img, row0 = (mt - img * tiles_per_img) * R;
  if (tid < BM) {
    const int ty = tid / W, tx = tid - ty * W;
    t_out[tid] = (uint32_t)(((img * p.h + row0 + ty) * W + tx) * p.ld_out) * 4u;
  }
  const int nchunks = p.ld_in / KCH;
  const int half = lane >> 5, col = lane & 31;
  f32x16 acc[2][4];

  if (wave >= 4) {
    // ================================================= loaders =====================================================================
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const int lt = tid - 256;
    const int qu = lt % UPR, row_t = lt / UPR;
    // rows advance by RPP (16 or 32) per pass: the swizzle term, bits 1..3 of the row, stays
    const int st_lds = BF16 ? lds_off(row_t, qu >> 1) + (qu & 1) * 8 : lds_off(row_t, qu);
    uint32_t a_voff[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int hp = row_t + RPP * i;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = row0 + hy - 1, ix = hx - 1;
      const bool ok = hp < HP && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)W;
      a_voff[i] = ok ? (uint32_t)(((img * p.h + iy) * W + ix) * p.ld_in + 4 * qu) * 4u : OOB;
    }
    const uint32_t b_voff = (uint32_t)(((int64_t)(n0 + row_t) * p.w_sn + 4 * qu) * 4);
    const uint32_t b_pass = (uint32_t)(RPP * p.w_sn * 4);
    u32x4 ra[A_IT], rb[TPS * B_IT];
    auto gload_a = [&](int c0) {
      const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_voff[i], so, 0);
    };
    auto put = [&](unsigned char* dst, u32x4 v) {
      if constexpr (BF16) *reinterpret_cast<u32x2*>(dst) = pack4(v);
      else *reinterpret_cast<u32x4*>(dst) = v;
    };
    auto sstore_a = [&]() {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (i + 1 < A_IT || row_t + RPP * i < HP) put(As + st_lds + i * RPP * 128, ra[i]);
    };
    auto gload_b = [&](int g, int c0) {
#pragma unroll
      for (int k = 0; k < TPS; ++k) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((p.tap[TPS * g + k] >> 16) * (int)p.w_st + c0) * 4);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) rb[k * B_IT + j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff, so + j * b_pass, 0);
      }
    };
    auto sstore_b = [&](int buf) {
      unsigned char* b = Bs + buf * B_BYTES + st_lds;
#pragma unroll
      for (int k = 0; k < TPS; ++k)
#pragma unroll
        for (int j = 0; j < B_IT; ++j) put(b + k * B_TAP + j * RPP * 128, rb[k * B_IT + j]);
    };
    gload_a(0);
    gload_b(0, 0);
    sstore_a();
    sstore_b(0);
    __syncthreads();                                            // (P) operands of the first step are in LDS
    int bbuf = 0;
    for (int c = 0; c < nchunks; ++c) {
      const bool more_c = c + 1 < nchunks;
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        const bool more = g < NG - 1 || more_c;
        if (more) {
          gload_b(g < NG - 1 ? g + 1 : 0, g < NG - 1 ? c * KCH : (c + 1) * KCH);
          if (g == 1 && more_c) gload_a((c + 1) * KCH);
          sstore_b(bbuf ^ 1);                                   // the other filter buffer: last read one step ago
        }
        __syncthreads();                                        // (S) the consumers are done with this step
        if (g == NG - 1 && more_c) {
          sstore_a();                                           // the ONE halo buffer: free now
          __syncthreads();                                      // (H)
        }
        bbuf ^= 1;
      }
    }
  } else {
    // ================================================= consumers ===================================================================
    const int wm0 = wave * 64;
    int a_hp[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int r = wm0 + mi * 32 + col;
      const int ty = r / W, tx = r - ty * W;
      a_hp[mi] = (ty + 1) * HW_ + tx + 1;
    }
    int b_off[4];                                               // filter row of fragment ni (its chunk is XOR-ed in below)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b_off[ni] = (ni * 32 + col) * 128;
    const int b_swz = (col >> 1) & 7;                           // rows ni*32 + col: bits 1..3 are col's
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    CSTAMP(1);
    __syncthreads();                                            // (P)
    CSTAMP(2);
    int bbuf = 0;
    for (int c = 0; c < nchunks; ++c) {
      const bool more_c = c + 1 < nchunks;
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        const unsigned char* B = Bs + bbuf * B_BYTES;
#pragma unroll
        for (int k = 0; k < TPS; ++k) {
          const int tp = p.tap[TPS * g + k];
          const int shift = (int)(int8_t)(tp >> 8) * HW_ + (int)(int8_t)tp;
          int a_row[2], a_swz[2];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const int hp = a_hp[mi] + shift;
            a_row[mi] = hp * 128;
            a_swz[mi] = (hp >> 1) & 7;
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {                          // 16-B chunk pair (2s, 2s+1) of the 128-B rows: lane half h takes chunk 2s + h
            if constexpr (BF16) {
              bf16x8 a[2], b[4];
#pragma unroll
              for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(B + k * B_TAP + b_off[ni] + (((2 * s + half) ^ b_swz) << 4));
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // D[pixel][channel]: lane = channel
            } else {
              f32x4 a[2], b[4];
#pragma unroll
              for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(B + k * B_TAP + b_off[ni] + (((2 * s + half) ^ b_swz) << 4));
#pragma unroll
              for (int e = 0; e < 4; ++e)                        // k = 8s + 4h + e for both operands
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                  for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
            }
          }
        }
        __syncthreads();                                        // (S)
        if (g == NG - 1 && more_c) __syncthreads();             // (H)
        bbuf ^= 1;
      }
    }
  }

  CSTAMP(3);
  // epilogue: as in the symmetric kernel — two passes of 128 rows through LDS, all 512 threads store
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  constexpr int TLD = BN + 4, PARTS = THREADS / BN;
  float* tile = reinterpret_cast<float*>(smem);
  float* red = tile + 128 * TLD;
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ymul ? p.ymul : p.out), 0, p.out_bytes, 0x00020000);
  int seg = 0;
  if (COLSUM) {
    const int m0 = mt * BM;
    int acc_rows = p.seg_rows[0];
    while (seg < p.nseg - 1 && m0 >= acc_rows) acc_rows += p.seg_rows[++seg];
  }
  const bool ym = COLSUM && p.ymul != nullptr;
  float csum = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    if (wave < 4 && (wave >> 1) == pass) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tile[((wave & 1) * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TLD + ni * 32 + col] = acc[mi][ni][r];
    }
    __syncthreads();
    constexpr int G4 = BN / 4;
    for (int i = tid; i < 128 * G4; i += THREADS) {
      const int rl = i / G4, cg = i - rl * G4;
      const int n = n0 + cg * 4;
      const uint32_t off = n >= p.n_store ? OOB : t_out[pass * 128 + rl] + (uint32_t)n * 4u;
      const float4 tv = *reinterpret_cast<const float4*>(tile + rl * TLD + cg * 4);
      float va[4] = {tv.x, tv.y, tv.z, tv.w};
      if (!COLSUM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (p.bias != nullptr && n + e < p.n_store) va[e] += p.bias[n + e];
          va[e] = tgd::act(va[e], p.act, p.alpha);
        }
      } else if (ym) {
        const u32x4 yb = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
        const uint32_t y0 = yb.x, y1 = yb.y, y2 = yb.z, y3 = yb.w;   // scalars first: see the symmetric kernel
        va[0] *= tgd::act_grad(__builtin_bit_cast(float, y0), p.ymul_act, p.ymul_alpha);
        va[1] *= tgd::act_grad(__builtin_bit_cast(float, y1), p.ymul_act, p.ymul_alpha);
        va[2] *= tgd::act_grad(__builtin_bit_cast(float, y2), p.ymul_act, p.ymul_alpha);
        va[3] *= tgd::act_grad(__builtin_bit_cast(float, y3), p.ymul_act, p.ymul_alpha);
        *reinterpret_cast<float4*>(tile + rl * TLD + cg * 4) = make_float4(va[0], va[1], va[2], va[3]);
      }
      if ((p.n_store & 3) == 0) {
        const u32x4 pk = {__builtin_bit_cast(uint32_t, va[0]), __builtin_bit_cast(uint32_t, va[1]), __builtin_bit_cast(uint32_t, va[2]),
                          __builtin_bit_cast(uint32_t, va[3])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_o, off, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, va[e]), rs_o, n + e >= p.n_store ? OOB : off + 4u * e, 0, 0);
      }
    }
    if (COLSUM) {
      if (ym) __syncthreads();
      {
        const int c = tid % BN, q = tid / BN;
        float s1 = 0.f;
        for (int rl = q; rl < 128; rl += PARTS) s1 += tile[rl * TLD + c];
        red[q * BN + c] = s1;
      }
      __syncthreads();
      if (tid < BN) {
#pragma unroll
        for (int q = 0; q < PARTS; ++q) csum += red[q * BN + tid];
      }
    }
    __syncthreads();
  }
  CSTAMP(4);
  if (COLSUM && tid < BN && n0 + tid < p.n_store) atomicAdd(p.colsum + (int64_t)seg * p.c_out + n0 + tid, (double)csum);
}

template <int W>
void launch_ws(ConvParams& p, hipStream_t s) {
  const dim3 grid(p.n_tiles_m * p.n_tiles_n);
  if (p.f32) {
    if (p.colsum) hipLaunchKernelGGL((conv3x3_ws_kernel<W, true, false>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_ws_kernel<W, false, false>), grid, dim3(512), 0, s, p);
  } else {
    if (p.colsum) hipLaunchKernelGGL((conv3x3_ws_kernel<W, true, true>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_ws_kernel<W, false, true>), grid, dim3(512), 0, s, p);
  }
}

// fp32 filter -> bf16, laid out as the consecutive LDS images the pipelined kernel's steps consume: image (nt, chunk, tap index k9 in the
// descriptor's tap order) = 128 filter rows x 64 channels = 16 KB with the 16-byte chunks XOR-swizzled exactly as lds_off() places them,
// so that a step's three images (48 KB) go global -> LDS by LDS-DMA as they stand (1 KB per wave-instruction, no registers, no
// conversion, no ds_write) and every workgroup fetches half the bytes from L2.  One thread per 16-byte chunk; ~150 k - 600 k elements.
__global__ void __launch_bounds__(256) filter_pack_kernel(ConvParams p, int nchunks, int total) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= total) return;
  const int q = u & 7, r = (u >> 3) & 127, img = u >> 10;
  const int k9 = img % 9, c = (img / 9) % nchunks, nt = img / (9 * nchunks);
  const float* src = p.w + (int64_t)(nt * BN + r) * p.w_sn + (int64_t)(p.tap[k9] >> 16) * p.w_st + c * KC + 8 * q;
  const u32x4 lo = *reinterpret_cast<const u32x4*>(src), hi = *reinterpret_cast<const u32x4*>(src + 4);
  const u32x2 a = pack4(lo), b = pack4(hi);
  u32x4 o;
  o.x = a.x; o.y = a.y; o.z = b.x; o.w = b.y;
  *reinterpret_cast<u32x4*>(static_cast<unsigned char*>(const_cast<void*>(p.wpk)) + (size_t)img * (BN * 128) + lds_off(r, q)) = o;
}

// s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier: the N youngest vector-memory operations of this wave stay in flight across the barrier
template <int N>
__device__ __forceinline__ void barrier_keep() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent, tile-pipelined form of the role-specialised kernel (the default).  Stamped on the form above (conv1_2, bf16 operands): of
// a workgroup's 59 k cycles 13 k are the prologue (first halo + filter group: HBM / L2 latency with the matrix pipe idle), 29 k the K
// loop and 17 k the epilogue (128 KB staged through LDS and stored by all eight waves) — one workgroup per CU, so nothing else runs
// on the CU meanwhile, and all 256 workgroups of a round are in the same phase at once (HBM idle during the K loops, matrix pipes idle
// during loads and stores).  Here a workgroup stays resident and walks its share of the tiles:
//   * the loaders treat (tile, chunk, tap group) as ONE stream — the first halo and filter group of the next tile are fetched during the
//     last steps of the current one, so only a workgroup's first tile pays a prologue;
//   * the consumers store their accumulators straight from registers (lane = channel: each half-wave writes one 128-B line of a pixel;
//     bias / activation, the activation-gradient multiplier and the column sums are per lane) — no LDS staging, no barrier, and the
//     stores drain while the next tile's MFMAs run;
//   * column sums: per wave in registers, one LDS slot per consumer wave, summed and added to the global accumulator (one double
//     atomic per channel and tile, as before) by the loader waves after the next barrier.
// Tile order: XCD x (= blockIdx & 7) owns a contiguous eighth of the tiles (neighbouring tiles share halo rows and all share the filter
// in that XCD's L2); its 32 workgroups stride through it.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int W, bool COLSUM, bool BF16>
__global__ void __launch_bounds__(512, 2) conv3x3_pipe_kernel(ConvParams p) {
  constexpr int BM = 256, TPS = 3, NG = 3;
  constexpr int KCH = BF16 ? 64 : 32;
  constexpr int UPR = BF16 ? 16 : 8;
  constexpr int RPP = 256 / UPR;
  constexpr int R = BM / W, HW_ = W + 2, HP = (R + 2) * HW_;
  constexpr int A_BYTES = (HP * 128 + 255) / 256 * 256, B_TAP = BN * 128, B_BYTES = TPS * B_TAP;
  constexpr int A_IT = (HP + RPP - 1) / RPP, B_IT = BN / RPP;
  constexpr int MAIN_BYTES = A_BYTES + 2 * B_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + 2 * 4 * BN * 4];
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;
  float* red = reinterpret_cast<float*>(smem + MAIN_BYTES);       // [tile parity][consumer wave][BN] column sums

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = p.n_tiles_m * p.n_tiles_n;
  int t, t_end, t_stride;
  if ((gridDim.x & 7) == 0) {
    const int x = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t = start + (blockIdx.x >> 3);
    t_end = start + q + (x < r ? 1 : 0);
    t_stride = gridDim.x >> 3;
  } else {
    t = blockIdx.x;
    t_end = ntiles;
    t_stride = gridDim.x;
  }
  if (t >= t_end) return;
  if (p.stagger > 0) {                                          // TG_CONV3X3_STAGGER (experiment): workgroup slot j of an XCD starts (j & 3) * stagger * ~1k cycles late
    const int d = ((blockIdx.x >> 3) & 3) * p.stagger;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(16);
  }
  const int nchunks = p.ld_in / KCH;
  const int tiles_per_img = p.h / R;
  const int half = lane >> 5, col = lane & 31;

  if (wave >= 4) {
    // ================================================= loaders =====================================================================
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const int lt = tid - 256;
    const int qu = lt % UPR, row_t = lt / UPR;
    const int st_lds = BF16 ? lds_off(row_t, qu >> 1) + (qu & 1) * 8 : lds_off(row_t, qu);
    uint32_t a_voff[A_IT];
    uint32_t b_voff;
    const uint32_t b_pass = (uint32_t)(RPP * p.w_sn * 4);
    auto set_tile_a = [&](int tt) {
      const int mt = tt / p.n_tiles_n;
      const int img = mt / tiles_per_img, row0 = (mt - img * tiles_per_img) * R;
      int rt = row_t;
      asm volatile("" : "+v"(rt));                              // keeps the per-pass halo coordinates from being hoisted out of the tile loop (22 x 2 registers)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int hp = rt + RPP * i;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = row0 + hy - 1, ix = hx - 1;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)W;
        a_voff[i] = ok ? (uint32_t)(((img * p.h + iy) * W + ix) * p.ld_in + 4 * qu) * 4u : OOB;
      }
    };
    int nt_b = 0, bbuf_w = 0;                                   // filter column tile the fetches read; filter buffer the next fetch fills
    auto set_tile_b = [&](int tt) {
      nt_b = tt % p.n_tiles_n;
      b_voff = (uint32_t)(((int64_t)(nt_b * BN + row_t) * p.w_sn + 4 * qu) * 4);
    };
    // The next halo waits in registers from step g = 1 (loads issued, after that step's filter group has been written, so that the two
    // register images are never live together) to the end of step g = 2; with bf16 operands it is packed at the top of step 2
    // (88 -> 44 registers) before that step's filter loads take their 96.
    u32x4 ra[A_IT], rb[BF16 ? 1 : TPS * B_IT];
    u32x2 rap[BF16 ? A_IT : 1];
    auto gload_a = [&](int c0) {
      const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (!(TG_ABL & 8) || i < A_IT / 2) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_voff[i], so, 0);
    };
    auto pack_a = [&]() {
      if constexpr (BF16) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) rap[i] = pack4(ra[i]);
      }
    };
    auto put = [&](unsigned char* dst, u32x4 v) {
      if constexpr (BF16) *reinterpret_cast<u32x2*>(dst) = pack4(v);
      else *reinterpret_cast<u32x4*>(dst) = v;
    };
    auto sstore_a = [&]() {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (i + 1 < A_IT || row_t + RPP * i < HP) {
          if constexpr (BF16) *reinterpret_cast<u32x2*>(As + st_lds + i * RPP * 128) = rap[i];
          else *reinterpret_cast<u32x4*>(As + st_lds + i * RPP * 128) = ra[i];
        }
    };
    // filter group (three taps) of the next step.  fp32 operands: global -> registers (gload_b) -> LDS (sstore_b).  bf16 operands: the
    // packed images go straight to LDS (gload_b: 12 LDS-DMA instructions of 1 KB per loader wave), nothing left for sstore_b.
    auto gload_b = [&](int g, int c0) {
      if constexpr (BF16) {
        const unsigned char* src = static_cast<const unsigned char*>(p.wpk) + ((size_t)((nt_b * nchunks + c0 / KCH) * NG + g)) * B_BYTES + (wave - 4) * (B_BYTES / 4) + lane * 16;
        unsigned char* dst = Bs + bbuf_w * B_BYTES + (wave - 4) * (B_BYTES / 4);
#pragma unroll
        for (int j = 0; j < B_BYTES / 4 / 1024; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024), (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                      // barrier_keep<A_IT> counts on the halo loads being issued AFTER these
      } else {
#pragma unroll
        for (int k = 0; k < TPS; ++k) {
          const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((p.tap[TPS * g + k] >> 16) * (int)p.w_st + c0) * 4);
#pragma unroll
          for (int j = 0; j < B_IT; ++j)
            if (!(TG_ABL & 16) || j < B_IT / 2) rb[k * B_IT + j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff, so + j * b_pass, 0);
        }
      }
    };
    auto sstore_b = [&](int buf) {
      if constexpr (!BF16) {
        unsigned char* b = Bs + buf * B_BYTES + st_lds;
#pragma unroll
        for (int k = 0; k < TPS; ++k)
#pragma unroll
          for (int j = 0; j < B_IT; ++j) put(b + k * B_TAP + j * RPP * 128, rb[k * B_IT + j]);
      }
    };
    auto flush = [&](int tt, int parity) {                      // column sums of tile tt: the four consumer slots -> one double atomic per channel
      if (lt < BN) {
        const int nt = tt % p.n_tiles_n, m0 = (tt / p.n_tiles_n) * BM;
        int seg = 0, acc_rows = p.seg_rows[0];
        while (seg < p.nseg - 1 && m0 >= acc_rows) acc_rows += p.seg_rows[++seg];
        const float* rp = red + parity * 4 * BN + lt;
        const float s1 = (rp[0] + rp[BN]) + (rp[2 * BN] + rp[3 * BN]);
        if (nt * BN + lt < p.n_store) atomicAdd(p.colsum + (int64_t)seg * p.c_out + nt * BN + lt, (double)s1);
      }
    };
    set_tile_a(t);
    set_tile_b(t);
    gload_b(0, 0);
    gload_a(0);
    pack_a();
    sstore_a();
    sstore_b(0);
    barrier_keep<0>();                                          // (P) operands of the first step are in LDS
    int bbuf = 0, k_tile = 0;
    while (true) {
      const int tn = t + t_stride;
      const bool has_next = tn < t_end;
      for (int c = 0; c < nchunks; ++c) {
        const bool last_c = c + 1 == nchunks;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool last_g = g == NG - 1;
          const bool more = !(last_g && last_c) || has_next;
          bool halo_in_flight = false;
          if (more) {
            if (last_g) pack_a();
            bbuf_w = bbuf ^ 1;                                  // the other filter buffer: last read one step ago
            if (!last_g) {
              gload_b(g + 1, c * KCH);
            } else if (!last_c) {
              gload_b(0, (c + 1) * KCH);
            } else {
              set_tile_b(tn);
              gload_b(0, 0);
            }
            sstore_b(bbuf ^ 1);
            if (g == 1) {
              if (!last_c) {
                gload_a((c + 1) * KCH);
                halo_in_flight = true;
              } else if (has_next) {
                set_tile_a(tn);
                gload_a(0);
                halo_in_flight = true;
              }
            }
          }
          // (S) the consumers are done with this step; the filter fetch has landed, the halo loads (issued after it) stay in flight
          if (halo_in_flight) barrier_keep<A_IT>();
          else barrier_keep<0>();
          if (COLSUM && c == 0 && g == 0 && k_tile > 0) flush(t - t_stride, (k_tile - 1) & 1);
          if (last_g && more) {
            sstore_a();                                         // the ONE halo buffer: free now
            barrier_keep<0>();                                  // (H)
          }
          bbuf ^= 1;
        }
      }
      if (!has_next) break;
      t = tn;
      ++k_tile;
    }
    if (COLSUM) {
      barrier_keep<0>();                                        // (F) the consumers' last column sums are in LDS
      flush(t, k_tile & 1);
    }
  } else {
    // ================================================= consumers ===================================================================
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ymul ? p.ymul : p.out), 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.w), 0, p.bias ? (uint32_t)p.n_store * 4u : 0u, 0x00020000);
    const bool ym = COLSUM && p.ymul != nullptr;
    const int wm0 = wave * 64;
    // Which pixel of its 32-pixel fragment lane `col` multiplies.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same + 32), conflict-free when the 16 halo rows of a group differ mod 16 (128-byte rows, the swizzle
    // key is row bits 1..3).  W = 32 / 64: the 32 pixels are consecutive in one halo row -> they do.  W = 16: lanes 16-31 sit in the
    // NEXT image row, 18 halo rows further on, and raster order makes rows 12, 13 mod 16 collide with the first row's (measured: 23 % of
    // the LDS cycles were conflicts); rotating the second row's pixels by two (lane 16 + i takes x = (i - 2) mod 16) restores
    // row = base + i mod 16.  The epilogue stores through the same map (frag_pixel).
    const int pc = (W == 16 && col >= 16) ? 16 + ((col - 18) & 15) : col;
    int a_hp[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int r = wm0 + mi * 32 + pc;
      const int ty = r / W, tx = r - ty * W;
      a_hp[mi] = (ty + 1) * HW_ + tx + 1;
    }
    int b_off[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b_off[ni] = (ni * 32 + col) * 128;
    const int b_swz = (col >> 1) & 7;
    f32x16 acc[2][4];
#ifdef TG_STAMP
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, sum_k = 0, sum_e = 0, sum_b = 0, tb0 = 0, tb1 = 0;
    LSTAMP(ts0);
#endif
    __syncthreads();                                            // (P)
#ifdef TG_STAMP
    LSTAMP(ts1);
#endif
    int bbuf = 0, k_tile = 0;
    while (true) {
      const int tn = t + t_stride;
      const bool has_next = tn < t_end;
#ifdef TG_STAMP
      LSTAMP(ts2);
#endif
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
      for (int c = 0; c < nchunks; ++c) {
        const bool last_c = c + 1 == nchunks;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
          const bool last_g = g == NG - 1;
          const bool more = !(last_g && last_c) || has_next;
          const unsigned char* B = Bs + bbuf * B_BYTES;
#pragma unroll
          for (int k = 0; k < TPS; ++k) {
            const int tp = p.tap[TPS * g + k];
            const int shift = (int)(int8_t)(tp >> 8) * HW_ + (int)(int8_t)tp;
            int a_row[2], a_swz[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              const int hp = a_hp[mi] + shift;
              a_row[mi] = hp * 128;
              a_swz[mi] = (hp >> 1) & 7;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              if constexpr (BF16) {
                bf16x8 a[2], b[4];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(B + k * B_TAP + b_off[ni] + (((2 * s + half) ^ b_swz) << 4));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                  for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // D[pixel][channel]: lane = channel
              } else {
                f32x4 a[2], b[4];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(B + k * B_TAP + b_off[ni] + (((2 * s + half) ^ b_swz) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
              }
            }
          }
#ifdef TG_STAMP
          LSTAMP(tb0);
#endif
          __syncthreads();                                      // (S)
          if (last_g && more) __syncthreads();                  // (H)
#ifdef TG_STAMP
          LSTAMP(tb1);
          sum_b += tb1 - tb0;
#endif
          bbuf ^= 1;
        }
      }
#ifdef TG_STAMP
      LSTAMP(ts3);
      sum_k += ts3 - ts2;
#endif
      // ---- epilogue of tile t: registers -> global.  Tiles are whole rows of whole images, so the tile's pixels are consecutive rows
      // of the [pixels][ld_out] output.  D[pixel][channel]: lane = channel ni*32 + col, register r of fragment mi = pixel
      // mi*32 + (r & 3) + 8*(r >> 2) + 4*half, so one dword store writes two whole 128-byte lines (one per half-wave).  Measured
      // (tools/micro/store_patterns.hip, one workgroup per CU storing such tiles): this shape and 16-byte stores of whole lines both
      // reach ~48 B/clk/CU when few CUs store and the chip-wide ~6 TB/s when all do; 16-byte stores straight from the transposed
      // accumulator (32 B per line and instruction) stay at 15 B/clk/CU.  Bias, the activation and the column sums are per lane.
      {
        const int n0 = (t % p.n_tiles_n) * BN;
        uint32_t px0 = (uint32_t)((t / p.n_tiles_n) * BM + wm0 + 4 * half);
        asm volatile("" : "+v"(px0));                             // per-tile value: keeps the 32 row offsets below out of the registers during the K loop
        const uint32_t row_bytes = (uint32_t)p.ld_out * 4u;
        // address = per-lane offset (this lane's channel in the wave's first pixel row; out of range beyond n_store) + a wave-uniform
        // scalar offset per (mi, r): four address registers instead of thirty-two
        uint32_t ch_off[4];
        float cs[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = n0 + ni * 32 + col;
          ch_off[ni] = n < p.n_store ? px0 * row_bytes + (uint32_t)n * 4u : OOB;
          cs[ni] = 0.f;
        }
        // pixel of fragment row (r & 3) + 8*(r >> 2) + 4*half, less the 4*half that px0 carries; W = 16: rows 16.. are rotated (see pc
        // above) and for r = 8, 9 the lower half-wave's pixels wrap around to the end of the image row (+16: wrap16)
        auto pix_off = [&](int mi, int r) {
          const int i = (W == 16 && r >= 8) ? 16 + 8 * ((r >> 2) - 2) + (r & 3) - 2 : (r & 3) + 8 * (r >> 2);
          return (uint32_t)(mi * 32 + i) * row_bytes;
        };
        const uint32_t wrap16 = (W == 16 && half == 0) ? 16u * row_bytes : 0u;
        auto lane_off = [&](int ni, int r) { return (W == 16 && (r == 8 || r == 9)) ? ch_off[ni] + wrap16 : ch_off[ni]; };
        if (!COLSUM) {
          // bias + activation; only y = x > 0 ? x : slope * x forms reach this kernel (none / relu / leaky relu: conv3x3_bf16_launch)
          const float slope = p.act == TG_ACT_LRELU ? p.alpha : (p.act == TG_ACT_RELU ? 0.f : 1.f);
          float bias_v[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const uint32_t bb = __builtin_amdgcn_raw_buffer_load_b32(rs_b, (uint32_t)(n0 + ni * 32 + col) * 4u, 0, 0);      // beyond n_store / no bias: 0
            bias_v[ni] = __builtin_bit_cast(float, bb);
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t off = pix_off(mi, r);
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) {
                const float x = acc[mi][ni][r] + bias_v[ni];
                const float v = x > 0.f ? x : slope * x;
                if (!(TG_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), off, 0);
              }
            }
        } else if (ym) {
          // the multiplier act'(y) is read at the output's own addresses, one (mi, ni) fragment (16 dwords per lane) ahead of its use
          const float slope = p.ymul_act == TG_ACT_LRELU ? p.ymul_alpha : (p.ymul_act == TG_ACT_RELU ? 0.f : 1.f);
          uint32_t yv[2][16];
          auto yload = [&](int f, int buf) {
            const int mi = f >> 2, ni = f & 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[buf][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_y, lane_off(ni, r), pix_off(mi, r), 0);
          };
          yload(0, 0);
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            const int mi = f >> 2, ni = f & 3;
            if (f + 1 < 8) yload(f + 1, (f + 1) & 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t yb = yv[f & 1][r];
              const float v = acc[mi][ni][r] * (__builtin_bit_cast(float, yb) > 0.f ? 1.f : slope);
              cs[ni] += v;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), pix_off(mi, r), 0);
            }
          }
        } else {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t off = pix_off(mi, r);
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) {
                const float v = acc[mi][ni][r];
                cs[ni] += v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), off, 0);
              }
            }
        }
        if (COLSUM) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float o = __shfl_xor(cs[ni], 32);
            if (half == 0) red[((k_tile & 1) * 4 + wave) * BN + ni * 32 + col] = cs[ni] + o;
          }
        }
      }
#ifdef TG_STAMP
      LSTAMP(ts2);
      sum_e += ts2 - ts3;
#endif
      if (!has_next) break;
      t = tn;
      ++k_tile;
    }
#ifdef TG_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 64) {                  // consumer wave 0: first barrier wait | K loops | of which at barriers | epilogues | total | tiles
      tg_conv_stamps[0 * 64 + blockIdx.x] = ts1 - ts0;
      tg_conv_stamps[1 * 64 + blockIdx.x] = sum_k;
      tg_conv_stamps[2 * 64 + blockIdx.x] = sum_b;
      tg_conv_stamps[3 * 64 + blockIdx.x] = sum_e;
      tg_conv_stamps[4 * 64 + blockIdx.x] = ts2 - ts0;
      tg_conv_stamps[5 * 64 + blockIdx.x] = (unsigned long long)(k_tile + 1);
    }
#endif
    if (COLSUM) __syncthreads();                                // (F)
  }
}

template <int W>
void launch_pipe(ConvParams& p, hipStream_t s) {
  const int tiles = p.n_tiles_m * p.n_tiles_n, cus = compute_units();
  const dim3 grid(tiles < cus ? tiles : cus);                  // one resident workgroup per CU (LDS), each walking its share of the tiles
  if (p.f32) {
    if (p.colsum) hipLaunchKernelGGL((conv3x3_pipe_kernel<W, true, false>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_pipe_kernel<W, false, false>), grid, dim3(512), 0, s, p);
  } else {
    if (p.colsum) hipLaunchKernelGGL((conv3x3_pipe_kernel<W, true, true>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_pipe_kernel<W, false, true>), grid, dim3(512), 0, s, p);
  }
}

template <int W, int BM, int TPS>
void launch(ConvParams& p, hipStream_t s) {
  const dim3 grid(p.n_tiles_m * p.n_tiles_n);
  if (p.colsum) hipLaunchKernelGGL((conv3x3_bf16_kernel<W, BM, TPS, true>), grid, dim3(2 * BM), 0, s, p);
  else hipLaunchKernelGGL((conv3x3_bf16_kernel<W, BM, TPS, false>), grid, dim3(2 * BM), 0, s, p);
}

// tile choice.  256 pixels x 3 taps per step: one workgroup per CU (141-150 KB of LDS), long K-steps; 128 pixels x 1 tap per step: two
// independent workgroups per CU (70 KB each) whose load / matrix / store phases interleave.  TG_CONV3X3_BM (read once) forces one.
int g_force_bm = 0, g_dbg = 0, g_stagger = 0;
const bool g_symmetric = getenv("TG_CONV3X3_SYMMETRIC") != nullptr;      // A/B: the symmetric (non role-specialised) 256-pixel kernel
const int g_env_loaded = ([] { if (const char* e = getenv("TG_CONV3X3_BM")) g_force_bm = atoi(e); if (const char* e = getenv("TG_CONV3X3_DBG")) g_dbg = atoi(e); if (const char* e = getenv("TG_CONV3X3_STAGGER")) g_stagger = atoi(e); return 0; })();
const bool g_staged = getenv("TG_CONV3X3_STAGED") != nullptr;            // A/B: one tile per workgroup, LDS-staged epilogue (conv3x3_ws_kernel)
const bool g_disabled = getenv("TG_NO_CONV3X3_BF16") != nullptr;     // A/B switches, read once at library load
const bool g_disabled_f32 = getenv("TG_NO_CONV3X3_F32") != nullptr;

// Scratch for the packed bf16 filters of conv3x3_pipe_kernel: a ring of four 8 MB device buffers, allocated at the first launch that wants one
// (never inside a stream capture — such a launch takes the staged kernel instead) and kept for the life of the process.  A launch packs
// into the next slot and the convolution that follows on the same stream reads it; launches of this path are issued from one host thread
// and are stream-ordered with each other (the package's single compute stream, or the linear graph captured from it), so a slot is
// rewritten only three convolutions after the one that read it.
constexpr size_t PACK_SLOT_BYTES = 8u << 20;
constexpr int PACK_SLOTS = 4;
void* g_pack[PACK_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
int g_pack_dev = -1, g_pack_next = 0;
bool g_pack_failed = false;

void* pack_slot(size_t need, hipStream_t s) {
  if (need > PACK_SLOT_BYTES || g_pack_failed) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (g_pack_dev < 0) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    for (int i = 0; i < PACK_SLOTS; ++i)
      if (hipMalloc(&g_pack[i], PACK_SLOT_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        for (int j = 0; j < i; ++j) (void)hipFree(g_pack[j]);
        g_pack_failed = true;
        return nullptr;
      }
    g_pack_dev = dev;
  }
  if (dev != g_pack_dev) return nullptr;
  void* r = g_pack[g_pack_next];
  g_pack_next = (g_pack_next + 1) % PACK_SLOTS;
  return r;
}

int g_policy = 0;                      // tg_conv3x3_policy: 0 = where it pays (below), 1 = wherever it applies, 2 = never
long g_launches = 0;

int pick_bm(const tg_igemm_desc* d) {
  const bool ok256 = d->h_in % (256 / d->w_in) == 0, ok128 = d->h_in % (128 / d->w_in) == 0;
  if (g_force_bm == 256 && ok256) return 256;
  if (g_force_bm == 128 && ok128) return 128;
  return ok256 ? 256 : (ok128 ? 128 : 0);
}

}  // namespace

namespace tg {

int halo_policy() { return g_policy; }
int halo_compute_units() { return compute_units(); }
void halo_count_launch() { ++g_launches; }

// the layer has the kernel's shape (independent of how many images the launch holds)
static bool conv3x3_fits(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  if ((bf16 ? g_disabled : g_disabled_f32) || g_policy == 2) return false;
  if (!bf16 && d->h_in % (256 / (d->w_in > 0 ? d->w_in : 1))) return false;          // the fp32 form exists for the 256-pixel tile only
  if (n_desc != 1 || d->n_taps != 9 || d->n_group != 0) return false;
  if (d->s_y != 1 || d->s_x != 1 || d->os_y != 1 || d->os_x != 1 || d->oo_y != 0 || d->oo_x != 0) return false;
  if (d->h_v != d->h_in || d->w_v != d->w_in || d->h_out != d->h_in || d->w_out != d->w_in) return false;
  if (d->w_in != 16 && d->w_in != 32 && d->w_in != 64) return false;
  if (pick_bm(d) == 0) return false;
  if (d->ld_in % (bf16 ? KC : 32) || d->c_out % BN) return false;
  bool seen[9] = {false, false, false, false, false, false, false, false, false};
  for (int t = 0; t < 9; ++t) {                               // the nine taps of a 3x3 window, each exactly once, in any order
    if (d->dy[t] < -1 || d->dy[t] > 1 || d->dx[t] < -1 || d->dx[t] > 1) return false;
    const int k = (d->dy[t] + 1) * 3 + d->dx[t] + 1;
    if (seen[k]) return false;
    seen[k] = true;
  }
  const int per_img = d->h_in * d->w_in;
  for (int i = 0; i < nseg; ++i)
    if (seg_rows[i] % per_img) return false;                  // applications are whole images: a tile never straddles a segment
  return true;
}

// One workgroup per CU (141-150 KB of LDS), so a launch runs in ceil(tiles / CUs) rounds and the last one may be nearly empty: 130 images
// of 32x32 are 520 tiles = 3 rounds on 256 CUs for 2.03 rounds of work.  The generic implicit GEMM (thousands of small workgroups) has no
// such step, so the halo form is taken only where its per-tile advantage (measured on full rounds: 1.07x with fp32 operands, 1.8x with
// bf16 ones) survives the quantisation.
static long conv3x3_slots(const tg_igemm_desc* d, bool bf16) { return (long)compute_units() * ((bf16 ? pick_bm(d) : 256) == 128 ? 2 : 1); }
static long conv3x3_tiles_per_image(const tg_igemm_desc* d, bool bf16) { return (long)d->h_in * d->w_in / (bf16 ? pick_bm(d) : 256) * (d->c_out / BN); }

static bool conv3x3_pays(const tg_igemm_desc* d, bool bf16, int n_img) {
  if (g_policy != 0) return true;
  const long slots = conv3x3_slots(d, bf16), tiles = conv3x3_tiles_per_image(d, bf16) * n_img;
  const long rounds = (tiles + slots - 1) / slots;
  return (double)tiles / (double)(rounds * slots) * (bf16 ? 1.8 : 1.07) >= 1.0;
}

bool conv3x3_bf16_applicable(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  return conv3x3_fits(d, n_desc, seg_rows, nseg, bf16) && conv3x3_pays(d, bf16, d->n_img);
}

// A launch of the right shape whose last round is less than 90 % full: the number of LEADING images that make whole rounds — they go to
// the halo kernel, the few images left to the generic one (igemm_impl splits the launch; 130 images of 32x32: 128 + 2).  0: no such
// split (rounds full enough, too few images for one round, or the whole-round prefix is less than 60 % of the launch).
int conv3x3_bf16_split_images(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  if (g_policy != 0 || !conv3x3_fits(d, n_desc, seg_rows, nseg, bf16)) return 0;
  const long slots = conv3x3_slots(d, bf16), tpi = conv3x3_tiles_per_image(d, bf16);
  const long tiles = tpi * d->n_img, rounds = (tiles + slots - 1) / slots;
  if (tiles * 10 >= rounds * slots * 9) return 0;
  const long full = tiles / slots;                            // whole rounds in the launch
  if (full < 1) return 0;
  const long head = full * slots / tpi;                       // images in them (rounded down: the head's last round may miss a few tiles)
  if (head < 1 || head >= d->n_img || head * 10 < (long)d->n_img * 6 || !conv3x3_pays(d, bf16, (int)head)) return 0;
  return (int)head;
}

int conv3x3_bf16_launch(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, double* colsum,
                        const int32_t* seg_rows, int nseg, const float* ymul, int ymul_act, float ymul_alpha, uint32_t in_bytes, uint32_t w_bytes,
                        uint32_t out_bytes, hipStream_t s, bool bf16) {
  ConvParams p;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.colsum = colsum; p.ymul = ymul; p.ymul_act = ymul_act; p.ymul_alpha = ymul_alpha;
  p.nseg = nseg;
  for (int i = 0; i < 8; ++i) p.seg_rows[i] = (seg_rows && i < nseg) ? seg_rows[i] : 0;
  p.n_img = d->n_img; p.h = d->h_in; p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.c_out = d->c_out; p.n_store = d->n_store;
  p.act = d->act; p.alpha = d->alpha;
  p.w_sn = d->w_sn; p.w_st = d->w_st;
  for (int t = 0; t < 9; ++t) p.tap[t] = ((int)d->tapw[t] << 16) | (((int)d->dy[t] & 0xff) << 8) | ((int)d->dx[t] & 0xff);
  const int bm = bf16 ? pick_bm(d) : 256;
  p.f32 = bf16 ? 0 : 1;
  p.dbg = g_dbg;
  p.stagger = g_stagger;
  p.n_tiles_m = d->n_img * d->h_in * d->w_in / bm;
  p.n_tiles_n = d->c_out / BN;
  p.in_bytes = in_bytes; p.w_bytes = w_bytes; p.out_bytes = out_bytes;
  ++g_launches;
  const auto simple = [](int a) { return a == TG_ACT_NONE || a == TG_ACT_LRELU || a == TG_ACT_RELU; };
  // measured (N = 250 images, conv1_2 / conv2_1 / conv2_2; staged -> pipelined): bf16 585 / 597 / 809 -> 869 / 739 / 961 TFLOP/s, with column sums
  // 616 / 632 / 852 -> 794 / 672 / 883, exact fp32 119 / 128 / 138 -> 123 / 132 / 141
  bool pipe = bm == 256 && !g_staged && (!g_symmetric || !bf16) && simple(d->act) && (ymul == nullptr || simple(ymul_act));
  p.wpk = nullptr;
  if (pipe && bf16) {
    const int nchunks = d->ld_in / KC;
    const int total = p.n_tiles_n * nchunks * 9 * 1024;        // 16-byte chunks of the packed filter
    p.wpk = pack_slot((size_t)total * 16, s);
    if (p.wpk == nullptr) pipe = false;
    else hipLaunchKernelGGL(filter_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, nchunks, total);
  }
  if (pipe) {
    if (d->w_in == 16) launch_pipe<16>(p, s);
    else if (d->w_in == 32) launch_pipe<32>(p, s);
    else launch_pipe<64>(p, s);
  } else if (bm == 256 && (!g_symmetric || !bf16)) {
    if (d->w_in == 16) launch_ws<16>(p, s);
    else if (d->w_in == 32) launch_ws<32>(p, s);
    else launch_ws<64>(p, s);
  } else if (bm == 256) {
    if (d->w_in == 16) launch<16, 256, 3>(p, s);
    else if (d->w_in == 32) launch<32, 256, 3>(p, s);
    else launch<64, 256, 3>(p, s);
  } else {
    if (d->w_in == 16) launch<16, 128, 1>(p, s);
    else if (d->w_in == 32) launch<32, 128, 1>(p, s);
    else launch<64, 128, 1>(p, s);
  }
  TG_CHECK_LAUNCH("conv3x3_bf16_kernel");
  return TG_OK;
}

}  // namespace tg

extern "C" int tg_conv3x3_policy(int policy) {
  const int was = g_policy;
  if (policy >= 0 && policy <= 2) g_policy = policy;
  return was;
}

extern "C" int64_t tg_conv3x3_launches(void) { return g_launches; }

#ifdef TG_STAMP
extern "C" int tg_debug_read_conv_stamps(unsigned long long* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_conv_stamps), sizeof(unsigned long long) * 8 * 64);
  return e == hipSuccess ? 0 : -2;
}
#endif
