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
//   * 8 waves (4 along pixels x 2 along channels, 64x64 outputs each = 2x2 v_mfma_f32_32x32x16_bf16 tiles), two per SIMD; the loads of
//     the next filter tile / next halo are issued before the 16 MFMAs of a K-step and written to the other LDS buffer after them; one
//     barrier per K-step (a K-step is 1024 matrix-pipe cycles per SIMD).
// Numerics: every MFMA operand is rounded to bf16 (RNE) exactly as the generic tg_*_bf16 kernels do, products accumulate in fp32 —
// the results differ from those kernels only by the order of the fp32 accumulation (channel chunks outermost here).
#include <cstdlib>
#include "tg_common.h"
#include "tg_device.h"
#include "tg_conv3x3_bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 256, BN = 128, KC = 64;     // pixels, output channels, channels per chunk
constexpr int THREADS = 512;
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
};

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

// LDS byte offset of 16-B chunk `chunk` (8 bf16) of row `row`: 128-B rows, chunk index XOR-swizzled with bits 1..3 of the row.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int W, bool COLSUM>
__global__ void __launch_bounds__(THREADS, 2) conv3x3_bf16_kernel(ConvParams p) {
  constexpr int R = BM / W;                       // image rows per tile
  constexpr int HW_ = W + 2, HP = (R + 2) * HW_;  // halo geometry
  constexpr int A_BYTES = (HP * 128 + 255) / 256 * 256, B_BYTES = BN * 128;
  constexpr int A_UNITS = HP * 16, A_IT = (A_UNITS + THREADS - 1) / THREADS;   // 16-B fp32 loads of one halo chunk, per thread
  constexpr int B_IT = BN * 16 / THREADS;                                      // = 4
  constexpr int EPI_BYTES = 128 * (BN + 4) * 4 + 2 * (THREADS / BN) * BN * 4;
  constexpr int MAIN_BYTES = 2 * A_BYTES + 2 * B_BYTES;
  constexpr int SMEM = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + BM * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * A_BYTES;
  uint32_t* t_out = reinterpret_cast<uint32_t*>(smem + (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, p.n_tiles_m * p.n_tiles_n);
  const int nt = lid % p.n_tiles_n, mt = lid / p.n_tiles_n;
  const int n0 = nt * BN;
  const int tiles_per_img = p.h / R;
  const int img = mt / tiles_per_img, row0 = (mt - img * tiles_per_img) * R;     // first image row of the tile

  if (tid < BM) {
    const int ty = tid / W, tx = tid - ty * W;
    t_out[tid] = (uint32_t)(((img * p.h + row0 + ty) * W + tx) * p.ld_out) * 4u;
  }

  // ---- staging addresses (fixed for the whole kernel; the channel chunk rides in the scalar offset of the buffer loads) ----------
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  uint32_t a_voff[A_IT];
  int a_lds[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int u = tid + i * THREADS;
    const int hp = u >> 4, q = u & 15;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int iy = row0 + hy - 1, ix = hx - 1;
    const bool ok = u < A_UNITS && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)W;
    a_voff[i] = ok ? (uint32_t)(((img * p.h + iy) * W + ix) * p.ld_in + 4 * q) * 4u : OOB;
    a_lds[i] = u < A_UNITS ? lds_off(hp, q >> 1) + (q & 1) * 8 : -1;
  }
  uint32_t b_voff[B_IT];
  int b_lds[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int u = tid + j * THREADS;
    const int n = u >> 4, q = u & 15;
    b_voff[j] = (uint32_t)(((int64_t)(n0 + n) * p.w_sn + 4 * q) * 4);
    b_lds[j] = lds_off(n, q >> 1) + (q & 1) * 8;
  }

  u32x4 ra[A_IT], rb[B_IT];
  auto gload_a = [&](int c0) {
    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_voff[i], so, 0);
  };
  auto sstore_a = [&](int buf) {
    unsigned char* a = As + buf * A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (a_lds[i] >= 0) *reinterpret_cast<u32x2*>(a + a_lds[i]) = pack4(ra[i]);
  };
  auto gload_b = [&](int t, int c0) {
    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((p.tap[t] >> 16) * (int)p.w_st + c0) * 4);
#pragma unroll
    for (int j = 0; j < B_IT; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff[j], so, 0);
  };
  auto sstore_b = [&](int buf) {
    unsigned char* b = Bs + buf * B_BYTES;
#pragma unroll
    for (int j = 0; j < B_IT; ++j) *reinterpret_cast<u32x2*>(b + b_lds[j]) = pack4(rb[j]);
  };

  // ---- fragment addresses --------------------------------------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;                  // 4 x 2 waves; each owns 64 pixels x 64 channels
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
  // prologue: halo of chunk 0, filter tile of (tap 0, chunk 0)
  gload_a(0);
  gload_b(0, 0);
  sstore_a(0);
  sstore_b(0);
  __syncthreads();

  int bbuf = 0;
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char* A = As + (c & 1) * A_BYTES;
    const bool more_c = c + 1 < nchunks;
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      const bool more = t < 8 || more_c;
      if (more) gload_b(t < 8 ? t + 1 : 0, t < 8 ? c * KC : (c + 1) * KC);
      if (t == 5 && more_c) gload_a((c + 1) * KC);          // next halo: issued three K-steps ahead, written behind the last tap
      const int tp = p.tap[t];
      const int shift = (int)(int8_t)(tp >> 8) * HW_ + (int)(int8_t)tp;
      const unsigned char* B = Bs + bbuf * B_BYTES;
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
        for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(A + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(B + b_off[ni][s]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // D[pixel][channel]: lane = channel
      }
      if (more) sstore_b(bbuf ^ 1);
      if (t == 8 && more_c) sstore_a((c + 1) & 1);
      __syncthreads();
      bbuf ^= 1;
    }
  }

  // ---- epilogue (the operand tiles are dead; t_out lies behind them) ---------------------------------------------------------------
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
  for (int pass = 0; pass < 2; ++pass) {
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
      const uint32_t off = n >= p.n_store ? OOB : t_out[pass * 128 + rl] + (uint32_t)n * 4u;
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
  if (COLSUM && tid < BN && n0 + tid < p.n_store) atomicAdd(p.colsum + (int64_t)seg * p.c_out + n0 + tid, (double)csum);
}

template <int W>
void launch(ConvParams& p, hipStream_t s) {
  const dim3 grid(p.n_tiles_m * p.n_tiles_n);
  if (p.colsum) hipLaunchKernelGGL((conv3x3_bf16_kernel<W, true>), grid, dim3(THREADS), 0, s, p);
  else hipLaunchKernelGGL((conv3x3_bf16_kernel<W, false>), grid, dim3(THREADS), 0, s, p);
}

}  // namespace

namespace tg {

bool conv3x3_bf16_applicable(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg) {
  if (getenv("TG_NO_CONV3X3_BF16")) return false;             // A/B switch (read per call: this is a host-side tuning aid of the bf16 opt-in path)
  if (n_desc != 1 || d->n_taps != 9 || d->n_group != 0) return false;
  if (d->s_y != 1 || d->s_x != 1 || d->os_y != 1 || d->os_x != 1 || d->oo_y != 0 || d->oo_x != 0) return false;
  if (d->h_v != d->h_in || d->w_v != d->w_in || d->h_out != d->h_in || d->w_out != d->w_in) return false;
  if (d->w_in != 16 && d->w_in != 32 && d->w_in != 64) return false;
  const int R = BM / d->w_in;
  if (d->h_in % R) return false;
  if (d->ld_in % KC || d->c_out % BN) return false;
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

int conv3x3_bf16_launch(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, double* colsum,
                        const int32_t* seg_rows, int nseg, const float* ymul, int ymul_act, float ymul_alpha, uint32_t in_bytes, uint32_t w_bytes,
                        uint32_t out_bytes, hipStream_t s) {
  ConvParams p;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.colsum = colsum; p.ymul = ymul; p.ymul_act = ymul_act; p.ymul_alpha = ymul_alpha;
  p.nseg = nseg;
  for (int i = 0; i < 8; ++i) p.seg_rows[i] = (seg_rows && i < nseg) ? seg_rows[i] : 0;
  p.n_img = d->n_img; p.h = d->h_in; p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.c_out = d->c_out; p.n_store = d->n_store;
  p.act = d->act; p.alpha = d->alpha;
  p.w_sn = d->w_sn; p.w_st = d->w_st;
  for (int t = 0; t < 9; ++t) p.tap[t] = ((int)d->tapw[t] << 16) | (((int)d->dy[t] & 0xff) << 8) | ((int)d->dx[t] & 0xff);
  p.n_tiles_m = d->n_img * d->h_in * d->w_in / BM;
  p.n_tiles_n = d->c_out / BN;
  p.in_bytes = in_bytes; p.w_bytes = w_bytes; p.out_bytes = out_bytes;
  if (d->w_in == 16) launch<16>(p, s);
  else if (d->w_in == 32) launch<32>(p, s);
  else launch<64>(p, s);
  TG_CHECK_LAUNCH("conv3x3_bf16_kernel");
  return TG_OK;
}

}  // namespace tg
