// MFMA implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate.
//
//   tg_igemm_f32 : out[p,n] = act(sum_{t,c} in[pix(p,t),c] * w[n,t,c] + bias[n])
//   tg_wgrad_f32 : slab[s][t][c][n] = sum_{p in split s} in[pix(p,t),c] * dout[p,n]
//
// One kernel family serves conv fwd (SAME/VALID, stride 1/2), conv input-gradient, 5x5 s2 transposed conv
// (one launch per output parity), dense / NiN / ZCA (1 tap) — the geometry lives in tg_igemm_desc.
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
//  * NHWC gathers are 16 B per lane, 128 B contiguous per 8 lanes; out-of-image taps load nothing.
#include "tg_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;    // reduction depth per LDS tile
constexpr int LDT = 36;   // padded LDS row stride (floats)

struct IgemmParams {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  tg_igemm_desc d;
  int M, n_tiles;
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

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256, 2) igemm_f32_kernel(IgemmParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;   // 16-B loads per thread per tile
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDT + 2 * BN * LDT + 4 * BM];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDT;
  int* t_base = reinterpret_cast<int*>(smem + 2 * BM * LDT + 2 * BN * LDT);
  int* t_y = t_base + BM;
  int* t_x = t_y + BM;
  int* t_out = t_x + BM;

  const tg_igemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockIdx.x % p.n_tiles, mt = blockIdx.x / p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;

  if (tid < BM) {
    int m = m0 + tid;
    int base = 0, y0 = -30000, x0 = 0, oo = -1;
    if (m < p.M) {
      int hw = d.h_v * d.w_v;
      int img = m / hw, rem = m - img * hw;
      int vy = rem / d.w_v, vx = rem - vy * d.w_v;
      base = img * d.h_in * d.w_in * d.ld_in;
      y0 = vy * d.s_y;
      x0 = vx * d.s_x;
      oo = ((img * d.h_out + vy * d.os_y + d.oo_y) * d.w_out + vx * d.os_x + d.oo_x) * d.ld_out;
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
  const float* wrow[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) wrow[j] = p.w + (int64_t)(n0 + lrow + 32 * j) * d.w_sn + seg * 4;

  f32x4 ra[AR], rb[BR];
  const int cchunks = d.ld_in / BK;
  const int nk = d.n_taps * cchunks;

  auto gload = [&](int tap, int c0) {
    const int dy = d.dy[tap], dx = d.dx[tap];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int iy = ay[j] + dy, ix = ax[j] + dx;
      bool ok = (unsigned)iy < (unsigned)d.h_in && (unsigned)ix < (unsigned)d.w_in;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.in + (abase[j] + (iy * d.w_in + ix) * d.ld_in + c0));
      ra[j] = v;
    }
    const int64_t wo = (int64_t)d.tapw[tap] * d.w_st + c0;
#pragma unroll
    for (int j = 0; j < BR; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wrow[j] + wo);
  };
  auto sstore = [&](int buf) {
    float* a = As + buf * BM * LDT + lrow * LDT + seg * 4;
    float* b = Bs + buf * BN * LDT + lrow * LDT + seg * 4;
#pragma unroll
    for (int j = 0; j < AR; ++j) *reinterpret_cast<f32x4*>(a + 32 * j * LDT) = ra[j];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(b + 32 * j * LDT) = rb[j];
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

  gload(0, 0);
  sstore(0);
  __syncthreads();

  int tap = 0, c0 = 0, buf = 0;
  for (int it = 0; it < nk; ++it) {
    const bool more = it + 1 < nk;
    if (more) {
      c0 += BK;
      if (c0 == d.ld_in) { c0 = 0; ++tap; }
      gload(tap, c0);
    }
    const float* A = As + buf * BM * LDT + wm0 * LDT + frag;
    const float* B = Bs + buf * BN * LDT + wn0 * LDT + frag;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(A + mi * 32 * LDT + g * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(B + ni * 32 * LDT + g * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: +bias, activation, masked store (lane = column, 16 rows per 32x32 tile)
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn0 + ni * 32 + col;
    const bool nok = n < d.n_store;
    const float bv = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oo = t_out[ml];
        if (oo >= 0 && nok) p.out[(int64_t)oo + n] = apply_act(acc[mi][ni][r] + bv, d.act, d.alpha);
      }
    }
  }
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
};

template <int CT, int NT, int WAVES_C, int WAVES_N, int WAVES_K>
__global__ void __launch_bounds__(256, 2) wgrad_f32_kernel(WgradParams p) {
  static_assert(WAVES_C * WAVES_N * WAVES_K == 4, "4 waves");
  constexpr int WC = CT / WAVES_C, WN = NT / WAVES_N, MI = WC / 32, NI = WN / 32;
  constexpr int AR = CT / 32, BR = NT / 32;            // 16-B loads per thread per tile
  constexpr int ASEG = CT / 4, BSEG = NT / 4;          // 16-B segments per pixel row
  constexpr int TILE = BK * (CT + NT);
  constexpr int RED = (WAVES_K > 1) ? (WAVES_K - 1) * 64 * 16 * MI * NI * WAVES_C * WAVES_N : 0;
  constexpr int SM = (2 * TILE > RED ? 2 * TILE : RED) + 4 * BK;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  int* tbl = reinterpret_cast<int*>(smem + (2 * TILE > RED ? 2 * TILE : RED));   // [2][2][BK]: in_off, out_off

  const tg_igemm_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int nt = b % p.n_tiles; b /= p.n_tiles;
  const int ct = b % p.c_tiles; b /= p.c_tiles;
  const int tap = b % d.n_taps;
  const int split = b / d.n_taps;
  const int c0 = ct * CT, n0 = nt * NT;
  const int p_begin = split * p.px_per_split;
  const int p_end = min(p.M, p_begin + p.px_per_split);
  const int nk = (p_end > p_begin) ? (p_end - p_begin + BK - 1) / BK : 0;
  const int dy = d.dy[tap], dx = d.dx[tap];

  auto fill_tbl = [&](int it) {          // threads < BK: offsets of pixel tile `it`
    int px = p_begin + it * BK + tid;
    int io = -1, oo = -1;
    if (px < p_end) {
      int hw = d.h_v * d.w_v;
      int img = px / hw, rem = px - img * hw;
      int vy = rem / d.w_v, vx = rem - vy * d.w_v;
      int iy = vy * d.s_y + dy, ix = vx * d.s_x + dx;
      if ((unsigned)iy < (unsigned)d.h_in && (unsigned)ix < (unsigned)d.w_in)
        io = ((img * d.h_in + iy) * d.w_in + ix) * d.ld_in;
      oo = ((img * d.h_out + vy * d.os_y + d.oo_y) * d.w_out + vx * d.os_x + d.oo_x) * d.ld_out;
    }
    tbl[(it & 1) * 2 * BK + tid] = io;
    tbl[(it & 1) * 2 * BK + BK + tid] = oo;
  };

  const int aseg = tid % ASEG, arow = tid / ASEG;      // rows advance by 256/ASEG per pass
  const int bseg = tid % BSEG, brow = tid / BSEG;
  f32x4 ra[AR], rb[BR];
  auto gload = [&](int it) {
    const int* ti = tbl + (it & 1) * 2 * BK;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int off = ti[arow + j * (256 / ASEG)];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (off >= 0) v = *reinterpret_cast<const f32x4*>(p.in + (off + c0 + aseg * 4));
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      int off = ti[BK + brow + j * (256 / BSEG)];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (off >= 0) v = *reinterpret_cast<const f32x4*>(p.dout + (off + n0 + bseg * 4));
      rb[j] = v;
    }
  };
  auto sstore = [&](int buf) {
    float* a = smem + buf * TILE;
    float* bb = a + BK * CT;
#pragma unroll
    for (int j = 0; j < AR; ++j) *reinterpret_cast<f32x4*>(a + (arow + j * (256 / ASEG)) * CT + aseg * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(bb + (brow + j * (256 / BSEG)) * NT + bseg * 4) = rb[j];
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

  if (nk > 0) {
    if (tid < BK) { fill_tbl(0); if (nk > 1) fill_tbl(1); }
    __syncthreads();
    gload(0);
    sstore(0);
    __syncthreads();
  }
  const int half = lane >> 5, col = lane & 31;
  int buf = 0;
  for (int it = 0; it < nk; ++it) {
    const bool more = it + 1 < nk;
    if (more) gload(it + 1);
    if (tid < BK && it + 2 < nk) fill_tbl(it + 2);   // table slot (it&1) was last read for tile `it`
    const float* A = smem + buf * TILE + (16 * half) * CT + wc0 + col;
    const float* B = smem + buf * TILE + BK * CT + (16 * half) * NT + wn0 + col;
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
    if (more) sstore(buf ^ 1);
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
          out[(int64_t)c * d.c_out + n] = acc[mi][ni][r];
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
  TG_REQUIRE((d->h_v - 1) * d->os_y + d->oo_y < d->h_out && (d->w_v - 1) * d->os_x + d->oo_x < d->w_out && d->oo_y >= 0 && d->oo_x >= 0,
             "igemm: virtual grid %dx%d (stride %d,%d offset %d,%d) exceeds output %dx%d", d->h_v, d->w_v, d->os_y, d->os_x,
             d->oo_y, d->oo_x, d->h_out, d->w_out);
  TG_REQUIRE((int64_t)d->n_img * d->h_in * d->w_in * d->ld_in < (1LL << 31) && (int64_t)d->n_img * d->h_out * d->w_out * d->ld_out < (1LL << 31),
             "igemm: tensor exceeds 2^31 elements");
  return TG_OK;
}

}  // namespace

extern "C" int tg_igemm_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, void* stream) {
  int rc = check_desc(d);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(in && w && out, "igemm: null buffer");
  IgemmParams p{in, w, bias, out, *d, 0, 0};
  p.M = d->n_img * d->h_v * d->w_v;
  const double flops = 2.0 * p.M * d->c_out * d->n_taps * d->ld_in;
  const double bytes = 4.0 * ((double)p.M * d->ld_in + (double)p.M * d->n_store + (double)d->c_out * d->n_taps * d->ld_in);
  tg::ProfScope prof(tg::PC_IGEMM, flops, bytes, tg::as_stream(stream));
  const int m_tiles = (p.M + 127) / 128;
  if (d->c_out % 128 == 0) {
    p.n_tiles = d->c_out / 128;
    hipLaunchKernelGGL((igemm_f32_kernel<128, 128, 2, 2>), dim3(m_tiles * p.n_tiles), dim3(256), 0, tg::as_stream(stream), p);
  } else if (d->c_out % 64 == 0) {
    p.n_tiles = d->c_out / 64;
    hipLaunchKernelGGL((igemm_f32_kernel<128, 64, 2, 2>), dim3(m_tiles * p.n_tiles), dim3(256), 0, tg::as_stream(stream), p);
  } else {
    p.n_tiles = d->c_out / 32;
    hipLaunchKernelGGL((igemm_f32_kernel<128, 32, 4, 1>), dim3(m_tiles * p.n_tiles), dim3(256), 0, tg::as_stream(stream), p);
  }
  TG_CHECK_LAUNCH("igemm_f32_kernel");
  return TG_OK;
}

template <int CT, int NT, int WC, int WN, int WK>
static void launch_wgrad(WgradParams& p, hipStream_t s) {
  p.c_tiles = p.d.ld_in / CT;
  p.n_tiles = p.d.c_out / NT;
  int blocks = p.n_split * p.d.n_taps * p.c_tiles * p.n_tiles;
  hipLaunchKernelGGL((wgrad_f32_kernel<CT, NT, WC, WN, WK>), dim3(blocks), dim3(256), 0, s, p);
}

extern "C" int tg_wgrad_f32(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream) {
  int rc = check_desc(d);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(in && dout && slab, "wgrad: null buffer");
  TG_REQUIRE(n_split >= 1, "wgrad: n_split=%d", n_split);
  TG_REQUIRE(d->c_out <= d->ld_out, "wgrad: c_out=%d exceeds ld_out=%d", d->c_out, d->ld_out);
  WgradParams p{in, dout, slab, *d, 0, n_split, 0, 0, 0};
  p.M = d->n_img * d->h_v * d->w_v;
  p.px_per_split = (((p.M + n_split - 1) / n_split) + BK - 1) / BK * BK;
  const double flops = 2.0 * p.M * d->c_out * d->n_taps * d->ld_in;
  const double bytes = 4.0 * ((double)p.M * d->ld_in + (double)p.M * d->c_out + (double)n_split * d->c_out * d->n_taps * d->ld_in);
  tg::ProfScope prof(tg::PC_WGRAD, flops, bytes, tg::as_stream(stream));
  hipStream_t s = tg::as_stream(stream);
  const int ct = d->ld_in % 128 == 0 ? 128 : (d->ld_in % 64 == 0 ? 64 : 32);
  const int nt = d->c_out % 128 == 0 ? 128 : (d->c_out % 64 == 0 ? 64 : 32);
  if (ct == 128 && nt == 128) launch_wgrad<128, 128, 2, 2, 1>(p, s);
  else if (ct == 128 && nt == 64) launch_wgrad<128, 64, 2, 2, 1>(p, s);
  else if (ct == 128 && nt == 32) launch_wgrad<128, 32, 4, 1, 1>(p, s);
  else if (ct == 64 && nt == 128) launch_wgrad<64, 128, 2, 2, 1>(p, s);
  else if (ct == 64 && nt == 64) launch_wgrad<64, 64, 2, 2, 1>(p, s);
  else if (ct == 64 && nt == 32) launch_wgrad<64, 32, 2, 1, 2>(p, s);
  else if (ct == 32 && nt == 128) launch_wgrad<32, 128, 1, 4, 1>(p, s);
  else if (ct == 32 && nt == 64) launch_wgrad<32, 64, 1, 2, 2>(p, s);
  else launch_wgrad<32, 32, 1, 1, 4>(p, s);
  TG_CHECK_LAUNCH("wgrad_f32_kernel");
  return TG_OK;
}
