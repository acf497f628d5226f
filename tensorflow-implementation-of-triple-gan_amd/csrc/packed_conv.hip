// 3x3 / stride 1 / SAME convolution of a FEW input channels (<= 16: the discriminators' first layer, 3 image + 10 label channels -> 32,
// Model/Good_GAN_cifar10.py:63-66, Model/Good_GAN.py:129-132; tf.layers.conv2d, Model/modle_base.py:157-168) as K-PACKED fp32 MFMA products:
//   y[n,i,j,co]      = act( sum_{ky,kx,ci} x[n, i+ky-1, j+kx-1, ci] * W[ky,kx,ci,co] + b[co] )           (tg_conv3x3_packed_fwd_f32)
//   dW[ky,kx,ci,co]  = sum_{n,i,j}         x[n, i+ky-1, j+kx-1, ci] * dy[n,i,j,co]                       (tg_conv3x3_packed_wgrad_f32)
// The generic implicit GEMM (igemm.hip) walks K in 32-channel chunks per tap: 13 channels cost 32, i.e. K = 288 for 117 products per
// output (round 4 timeline: 0.13 ms of forward launches and the 0.08 ms filter gradient that ends the D-update, at 30 TFLOP/s of useful
// work).  Here the contraction index is the (tap, channel) PAIR itself, K = 9 c_in rounded up to the MFMA's k-pair: an operand element
// (pixel, k) is read from the input patch in LDS at pixel offset + ktab[k], the filter is the variable's own HWIO rows (no preparation
// launch), bias, activation and — for the discriminators' conv -> concat pairs — the label channels and the channel padding of the wider
// buffer are written by the same launch (as tg_igemm_labels_*).  Exact fp32 products (v_mfma_f32_32x32x2_f32), fp32 accumulation: the
// results differ from the generic kernel's only by the order of the sum.
#include "tg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_CIN = 16;
constexpr int FROWS = 8;             // forward block: up to 8 image rows x 16 pixels = four 32-pixel row pairs, one per wave
constexpr int PCOLS = 18;            // its input patch: 10 rows x 18 columns

__host__ __device__ inline int kpad(int c_in) { return (9 * c_in + 1) & ~1; }

template <int NT>
__global__ void __launch_bounds__(256) packed_fwd(const float* __restrict__ x, int ld_x, int c_in, const float* __restrict__ wk, const float* __restrict__ bias,
                                                  float slope, const float* __restrict__ lab, int lab_n, float* __restrict__ y, int ld_y, int h, int w) {
  constexpr int CO = 32 * NT, PR = FROWS + 2;
  extern __shared__ float lds[];
  const int K = 9 * c_in, KP = kpad(c_in);
  float* Wl = lds;                                   // [KP][CO]: the HWIO variable as it stands, zero rows beyond K
  float* Pl = lds + KP * CO;                         // [PR][PCOLS][c_in] input patch: rows i0-1 .. i0+8, columns j0-1 .. j0+16
  int* ktab = reinterpret_cast<int*>(Pl + PR * PCOLS * c_in);
  const int tiles_x = w / 16, tiles_y = (h + FROWS - 1) / FROWS;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int i0 = ty * FROWS, j0 = tx * 16;
  const int tid = threadIdx.x;
  // staging in groups of eight loads per thread: issued together, stored together (one load per loop trip made a workgroup's prologue a chain
  // of ~25 memory latencies: 48 us per 250-image launch, most of it here)
  for (int e0 = tid; e0 < KP * CO; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      v[u] = e < K * CO ? wk[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e < KP * CO) Wl[e] = v[u];
    }
  }
  {
    const int ci = tid & 15;
    constexpr int NP = (PR * PCOLS + 15) / 16;       // 12 passes of 16 patch pixels
    float v[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int p = (tid >> 4) + 16 * u;
      const int r = p / PCOLS, c = p - r * PCOLS;
      const int iy = i0 - 1 + r, ix = j0 - 1 + c;
      v[u] = (p < PR * PCOLS && ci < c_in && (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? x[(((int64_t)n * h + iy) * w + ix) * ld_x + ci] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int p = (tid >> 4) + 16 * u;
      if (p < PR * PCOLS && ci < c_in) Pl[p * c_in + ci] = v[u];
    }
  }
  if (tid < KP) {
    const int ky = tid / (3 * c_in), rem = tid - ky * 3 * c_in, kx = rem / c_in, ci = rem - kx * c_in;
    ktab[tid] = tid < K ? (ky * PCOLS + kx) * c_in + ci : 0;     // k >= K multiplies a zero filter row
  }
  __syncthreads();
  const int wv = tid >> 6, l = tid & 63, m = l & 31, hl = l >> 5;
  const int rows_blk = min(FROWS, h - i0);
  if (2 * wv >= rows_blk) return;
  const float* pa = Pl + ((2 * wv + (m >> 4)) * PCOLS + (m & 15)) * c_in;      // A[pixel m][k] = patch[pixel + ktab[k]]
  const float* pb = Wl + hl * CO + m;                                         // B[k = 2s + hl][co = 32q + m]
  f32x16 acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  int ko = ktab[hl];
  for (int s = 0; s < KP / 2; ++s) {
    const float a = pa[ko];
    if (s + 1 < KP / 2) ko = ktab[2 * (s + 1) + hl];
    const float* wb = pb + 2 * s * CO;
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[32 * q], acc[q], 0, 0, 0);    // D[pixel][co]: lane = channel
  }
  float bv[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) bv[q] = bias ? bias[32 * q + m] : 0.f;
  const int extra = ld_y - CO;                        // channels behind the convolution's: the label vector of the image, then zeros
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mo = (r & 3) + 8 * (r >> 2) + 4 * hl;
    const int i = i0 + 2 * wv + (mo >> 4), j = j0 + (mo & 15);
    float* o = y + (((int64_t)n * h + i) * w + j) * ld_y + m;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const float v = acc[q][r] + bv[q];
      o[32 * q] = v > 0.f ? v : slope * v;
    }
    for (int c = m; c < extra; c += 32) o[CO + c - m] = (lab && c < lab_n) ? lab[(int64_t)n * lab_n + c] : 0.f;
  }
}

// ---- filter gradient: dW[k][co] = sum_pixels P[pixel][k] * dy[pixel][co];  block = RBW image rows of one image, the (k tile, column tile)
// pairs dealt to the four waves; the three patch rows and the gradient row of one image row in LDS, the next on their way in registers ----
template <int NT>
__global__ void __launch_bounds__(256, 2) packed_wgrad(const float* __restrict__ x, int ld_x, int c_in, const float* __restrict__ dy, int ld_dy, int h, int w,
                                                      int rbw, float* __restrict__ part) {
  constexpr int CO = 32 * NT, NMT_MAX = (9 * MAX_CIN + 31) / 32, PERW = (NMT_MAX * NT + 3) / 4;
  constexpr int XP = 7, DU = 2 * NT;                   // staging registers: 3 * (w + 2) <= 102 patch pixels / 16 per pass; w * CO / 4 <= 512 NT units / 256
  extern __shared__ float lds[];
  const int K = 9 * c_in, KP = kpad(c_in), nmt = (KP + 31) / 32, n_tiles = nmt * NT, PW = w + 2;
  float* Xl = lds;                                     // [3][PW][c_in] input rows i-1 .. i+1
  float* Dl = lds + 3 * PW * c_in;                     // [w][CO] gradient row i
  const int tiles_y = h / rbw;
  const int ty = blockIdx.x % tiles_y, n = blockIdx.x / tiles_y;
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, m = l & 31, hl = l >> 5;
  int a_off[PERW], b_off[PERW];
#pragma unroll
  for (int a = 0; a < PERW; ++a) {
    const int t = wv + 4 * a, mt = t / NT, nt = t - mt * NT;
    const int k = 32 * mt + m;
    const int ky = k / (3 * c_in), rem = k - ky * 3 * c_in, kx = rem / c_in, ci = rem - kx * c_in;
    a_off[a] = (k < K ? (ky * PW + kx) * c_in + ci : 0) + hl * c_in;          // rows k >= K are computed on some in-range element and never stored
    b_off[a] = hl * CO + 32 * nt + m;
  }
  f32x16 acc[PERW];
#pragma unroll
  for (int a = 0; a < PERW; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int ci_l = tid & 15, pp = tid >> 4, npix = 3 * PW, dunits = w * (CO / 4);
  float xr[XP];
  f32x4 dr[DU];
  auto gload = [&](int i) {
#pragma unroll
    for (int u = 0; u < XP; ++u) {
      const int p = pp + 16 * u;
      const int r = p / PW, c = p - r * PW;
      const int iy = i - 1 + r, ix = c - 1;
      xr[u] = (p < npix && ci_l < c_in && (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? x[(((int64_t)n * h + iy) * w + ix) * ld_x + ci_l] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int e = tid + 256 * u < dunits ? tid + 256 * u : 0;
      const int j = e / (CO / 4), c4 = e - j * (CO / 4);
      dr[u] = *reinterpret_cast<const f32x4*>(dy + (((int64_t)n * h + i) * w + j) * ld_dy + 4 * c4);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int u = 0; u < XP; ++u) {
      const int p = pp + 16 * u;
      if (p < npix && ci_l < c_in) Xl[p * c_in + ci_l] = xr[u];
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int e = tid + 256 * u;
      if (e < dunits) *reinterpret_cast<f32x4*>(Dl + 4 * e) = dr[u];
    }
  };
  const int i_first = ty * rbw;
  gload(i_first);
  for (int i = i_first; i < i_first + rbw; ++i) {
    __syncthreads();                                   // the previous row's fragments have been read
    sstore();
    __syncthreads();
    if (i + 1 < i_first + rbw) gload(i + 1);           // in flight while this row is multiplied
    int xo = 0;
    for (int s = 0; s < w / 2; ++s, xo += 2 * c_in) {  // pixel pair (2s, 2s + 1): lane half hl takes pixel 2s + hl
#pragma unroll
      for (int a = 0; a < PERW; ++a)
        if (wv + 4 * a < n_tiles)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xl[a_off[a] + xo], Dl[b_off[a] + 2 * s * CO], acc[a], 0, 0, 0);    // D[k][co]
    }
  }
  float* o = part + (int64_t)blockIdx.x * K * CO;
#pragma unroll
  for (int a = 0; a < PERW; ++a) {
    const int t = wv + 4 * a;
    if (t < n_tiles) {
      const int mt = t / NT, nt = t - mt * NT;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hl;
        if (k < K) o[k * CO + 32 * nt + m] = acc[a][r];
      }
    }
  }
}

// dw[e] = sum over the blocks' partials in a fixed order: 64 outputs per workgroup, sixteen threads per output each summing every sixteenth
// partial (two accumulators), combined through LDS in a fixed tree
__global__ void __launch_bounds__(1024) packed_wgrad_reduce(const float* __restrict__ part, int n_part, int total, float* __restrict__ dw) {
  __shared__ float red[1024];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (e < total) {
    int k = kg;
    for (; k + 16 < n_part; k += 32) {
      s0 += part[(int64_t)k * total + e];
      s1 += part[(int64_t)(k + 16) * total + e];
    }
    if (k < n_part) s0 += part[(int64_t)k * total + e];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (kg < 4) red[threadIdx.x] = (red[threadIdx.x] + red[threadIdx.x + 256]) + (red[threadIdx.x + 512] + red[threadIdx.x + 768]);
  __syncthreads();
  if (kg == 0 && e < total) dw[e] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

size_t fwd_lds_bytes(int c_in, int c_out) { return (size_t)(kpad(c_in) * c_out + (FROWS + 2) * PCOLS * c_in + kpad(c_in)) * 4; }
size_t wgrad_lds_bytes(int w, int c_in, int c_out) { return (size_t)(3 * (w + 2) * c_in + w * c_out) * 4; }
int wgrad_rows(int h) { return h % 8 == 0 ? 8 : 4; }

bool shape_ok(int n, int h, int w, int c_in, int c_out) {
  return n > 0 && h > 0 && w > 0 && c_in >= 1 && c_in <= MAX_CIN && (c_out == 32 || c_out == 64) && w % 16 == 0 && w <= 32 && h % 4 == 0 &&
         fwd_lds_bytes(c_in, c_out) <= 64 * 1024 && wgrad_lds_bytes(w, c_in, c_out) <= 64 * 1024;
}

}  // namespace

extern "C" int tg_conv3x3_packed_supported(int n, int h, int w, int c_in, int c_out) { return shape_ok(n, h, w, c_in, c_out) ? 1 : 0; }

extern "C" int64_t tg_conv3x3_packed_wgrad_workspace_bytes(int n, int h, int w, int c_in, int c_out) {
  if (!shape_ok(n, h, w, c_in, c_out)) { tg::set_error("conv3x3_packed: shape not supported"); return TG_ERR_INVALID; }
  return (int64_t)n * (h / wgrad_rows(h)) * 9 * c_in * c_out * 4;
}

extern "C" int tg_conv3x3_packed_fwd_f32(const float* x, int ld_x, int c_in, const float* kernel, const float* bias, int act, float alpha, const float* labels,
                                         int lab_n, float* y, int ld_y, int n, int h, int w, int c_out, void* stream) {
  TG_REQUIRE(x && kernel && y, "conv3x3_packed_fwd: null buffer");
  TG_REQUIRE(shape_ok(n, h, w, c_in, c_out) && ld_x >= c_in && ld_y >= c_out && lab_n >= 0 && (!labels || c_out + lab_n <= ld_y),
             "conv3x3_packed_fwd: unsupported shape n=%d h=%d w=%d c_in=%d c_out=%d ld_x=%d ld_y=%d labels=%d", n, h, w, c_in, c_out, ld_x, ld_y, lab_n);
  TG_REQUIRE(act == TG_ACT_NONE || act == TG_ACT_RELU || act == TG_ACT_LRELU, "conv3x3_packed_fwd: activation %d is not y = x > 0 ? x : slope * x", act);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_IGEMM, 2.0 * n * h * w * kpad(c_in) * c_out, 4.0 * ((double)n * h * w * (c_in + ld_y)), s, "packed 3x3 conv fwd");
  const float slope = act == TG_ACT_LRELU ? alpha : (act == TG_ACT_RELU ? 0.f : 1.f);
  const dim3 grid(n * ((h + FROWS - 1) / FROWS) * (w / 16));
  const size_t sh = fwd_lds_bytes(c_in, c_out);
  if (c_out == 32) hipLaunchKernelGGL(packed_fwd<1>, grid, dim3(256), sh, s, x, ld_x, c_in, kernel, bias, slope, labels, lab_n, y, ld_y, h, w);
  else hipLaunchKernelGGL(packed_fwd<2>, grid, dim3(256), sh, s, x, ld_x, c_in, kernel, bias, slope, labels, lab_n, y, ld_y, h, w);
  TG_CHECK_LAUNCH("packed_fwd");
  return TG_OK;
}

extern "C" int tg_conv3x3_packed_wgrad_f32(const float* x, int ld_x, int c_in, const float* dy, int ld_dy, int n, int h, int w, int c_out, float* workspace,
                                           float* dw, void* stream) {
  TG_REQUIRE(x && dy && workspace && dw, "conv3x3_packed_wgrad: null buffer");
  TG_REQUIRE(shape_ok(n, h, w, c_in, c_out) && ld_x >= c_in && ld_dy >= c_out && ld_dy % 4 == 0,
             "conv3x3_packed_wgrad: unsupported shape n=%d h=%d w=%d c_in=%d c_out=%d ld_x=%d ld_dy=%d", n, h, w, c_in, c_out, ld_x, ld_dy);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_WGRAD, 2.0 * n * h * w * kpad(c_in) * c_out, 4.0 * ((double)n * h * w * (c_in + c_out)), s, "packed 3x3 conv wgrad");
  const int rbw = wgrad_rows(h), blocks = n * (h / rbw);
  const size_t sh = wgrad_lds_bytes(w, c_in, c_out);
  if (c_out == 32) hipLaunchKernelGGL(packed_wgrad<1>, dim3(blocks), dim3(256), sh, s, x, ld_x, c_in, dy, ld_dy, h, w, rbw, workspace);
  else hipLaunchKernelGGL(packed_wgrad<2>, dim3(blocks), dim3(256), sh, s, x, ld_x, c_in, dy, ld_dy, h, w, rbw, workspace);
  TG_CHECK_LAUNCH("packed_wgrad");
  const int total = 9 * c_in * c_out;
  hipLaunchKernelGGL(packed_wgrad_reduce, dim3((total + 63) / 64), dim3(1024), 0, s, workspace, blocks, total, dw);
  TG_CHECK_LAUNCH("packed_wgrad_reduce");
  return TG_OK;
}
