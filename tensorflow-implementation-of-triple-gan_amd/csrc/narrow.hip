// Backward pass of a 5x5 / stride-2 / 'same' transposed convolution with a NARROW output (the generator's image layer: 138 -> 3 channels,
// Model/Good_GAN_cifar10.py:55-57, Model/Good_GAN.py:76-83; tf.layers.conv2d_transpose, Model/modle_base.py:246-259) on the vector ALUs.
//
// Why not the MFMA kernels: their tiles pad the 3 image channels to 32, so the input gradient (K = 25 taps x 32 "channels") and the filter
// gradient (32 x 160 outputs per tap) do ten times the arithmetic the layer has, at a third of the matrix rate (round 2: 0.094 + 0.138 ms
// per step for 0.27 + 0.61 GFLOP of real work).  With c_out <= 4 the layer is small enough for plain FMAs:
//   dx[n,i,j,ci]     = sum_{ky,kx,co} dy[n, 2i+ky-1, 2j+kx-1, co] * W[ky,kx,co,ci]          (tg_deconv5x5s2_narrow_dgrad_f32)
//   dW[ky,kx,co,ci]  = sum_{n,i,j}    dy[n, 2i+ky-1, 2j+kx-1, co] * x[n,i,j,ci]            (tg_deconv5x5s2_narrow_wgrad_f32)
// (out-of-image dy positions contribute nothing; 'same' padding of the forward op: out[2i+ky-1, 2j+kx-1] += x[i,j] W[ky,kx]).
// Both kernels stage the dy patch of their pixel block in LDS and keep lane = input channel, so every global access is a coalesced NHWC row.
#include "tg_common.h"

namespace {

constexpr int TAPS = 25;
constexpr int MAXQ = 8;              // input channels per lane: ci_p <= 256

// ---- input gradient: block = 2 image rows x 16 pixels, thread = (channel lane, pixel group of 4) ----------------------------------------
template <int CO>
__global__ void __launch_bounds__(256) narrow_dgrad(const float* __restrict__ dy, int ld_dy, const float* __restrict__ kernel, const float* __restrict__ scale_a,
                                                    int c_in, int h, int w, int ci_p, float* __restrict__ dx, int ld_dx) {
  extern __shared__ float lds[];
  float* Wl = lds;                                   // [25 * CO][ci_p]
  float* Pl = lds + TAPS * CO * ci_p;                // [7][35][CO] dy patch: rows 2*i0-1 .. 2*i0+5, columns 2*j0-1 .. 2*j0+33
  const int tiles_x = w / 16, tiles_y = h / 2;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int i0 = ty * 2, j0 = tx * 16;
  const int tid = threadIdx.x, lane = tid & 31, pg = tid >> 5;
  for (int e = tid; e < TAPS * CO * ci_p; e += 256) {            // W[(tap, co)][ci]: the variable's own [5,5,Cout,Cin] rows (coalesced), channel-padded with zeros
    const int ci = e % ci_p, tc = e / ci_p;
    Wl[e] = ci < c_in ? kernel[(int64_t)tc * c_in + ci] * (scale_a ? scale_a[tc % CO] : 1.f) : 0.f;
  }
  const int H2 = 2 * h, W2 = 2 * w;
  for (int e = tid; e < 7 * 35 * CO; e += 256) {
    const int co = e % CO, c = (e / CO) % 35, r = e / (CO * 35);
    const int oy = 2 * i0 - 1 + r, ox = 2 * j0 - 1 + c;
    Pl[e] = ((unsigned)oy < (unsigned)H2 && (unsigned)ox < (unsigned)W2) ? dy[(((int64_t)n * H2 + oy) * W2 + ox) * ld_dy + co] : 0.f;
  }
  __syncthreads();
  const int nq = ci_p / 32;
  float acc[4][MAXQ];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[p][q] = 0.f;
  for (int ky = 0; ky < 5; ++ky)
    for (int kx = 0; kx < 5; ++kx)
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float* wr = Wl + ((ky * 5 + kx) * CO + co) * ci_p + lane;
        float d[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {                            // pixel p of this thread: row i0 + (p >> 1), column j0 + pg + 8 (p & 1)
          const int r = 2 * (p >> 1) + ky, c = 2 * (pg + 8 * (p & 1)) + kx;
          d[p] = Pl[(r * 35 + c) * CO + co];
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q)
          if (q < nq) {
            const float wv = wr[32 * q];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p][q] += d[p] * wv;
          }
      }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int i = i0 + (p >> 1), j = j0 + pg + 8 * (p & 1);
    float* o = dx + (((int64_t)n * h + i) * w + j) * ld_dx + lane;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
      if (q < nq) o[32 * q] = acc[p][q];
  }
}

// ---- filter gradient: block = RB image rows of one image, thread = (channel lane, (tap, co) group) --------------------------------------
constexpr int RB = 4;                                  // image rows per block
template <int CO>
__global__ void __launch_bounds__(256) narrow_wgrad(const float* __restrict__ dy, int ld_dy, const float* __restrict__ x, int ld_x, int h, int w, int ci_p,
                                                    float* __restrict__ part) {
  extern __shared__ float lds[];
  float* Xl = lds;                                     // [w][ci_p] one image row of x
  float* Pl = lds + w * ci_p;                          // [5][2w + 3][CO] the dy rows 2i-1 .. 2i+3
  const int tiles_y = h / RB;
  const int ty = blockIdx.x % tiles_y, n = blockIdx.x / tiles_y;
  const int tid = threadIdx.x, lane = tid & 31, g = tid >> 5;
  const int nq = ci_p / 32, PW = 2 * w + 3, H2 = 2 * h, W2 = 2 * w;
  constexpr int NTC = TAPS * CO, PER = (NTC + 7) / 8;  // (tap, co) pairs of group g: g, g + 8, ...
  float acc[PER][MAXQ];
  int poff[PER];                                       // patch offset of this thread's a-th (tap, co) pair at pixel column 0; -1: none
#pragma unroll
  for (int a = 0; a < PER; ++a) {
    const int tc = g + 8 * a;
    const int tap = tc / CO, co = tc - tap * CO, ky = tap / 5, kx = tap - ky * 5;
    poff[a] = tc < NTC ? (ky * PW + kx) * CO + co : -1;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[a][q] = 0.f;
  }
  for (int i = ty * RB; i < ty * RB + RB; ++i) {
    __syncthreads();
    for (int e = tid; e < w * ci_p; e += 256) {
      const int ci = e % ci_p, j = e / ci_p;
      Xl[e] = x[(((int64_t)n * h + i) * w + j) * ld_x + ci];
    }
    for (int e = tid; e < 5 * PW * CO; e += 256) {
      const int co = e % CO, c = (e / CO) % PW, r = e / (CO * PW);
      const int oy = 2 * i - 1 + r, ox = c - 1;
      Pl[e] = ((unsigned)oy < (unsigned)H2 && (unsigned)ox < (unsigned)W2) ? dy[(((int64_t)n * H2 + oy) * W2 + ox) * ld_dy + co] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < w; ++j) {
      float xv[MAXQ];
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) xv[q] = q < nq ? Xl[j * ci_p + lane + 32 * q] : 0.f;
#pragma unroll
      for (int a = 0; a < PER; ++a) {
        if (poff[a] >= 0) {
          const float d = Pl[poff[a] + 2 * j * CO];
#pragma unroll
          for (int q = 0; q < MAXQ; ++q) acc[a][q] += d * xv[q];
        }
      }
    }
  }
  float* o = part + (int64_t)blockIdx.x * NTC * ci_p;
#pragma unroll
  for (int a = 0; a < PER; ++a) {
    const int tc = g + 8 * a;
    if (tc < NTC)
#pragma unroll
      for (int q = 0; q < MAXQ; ++q)
        if (q < nq) o[tc * ci_p + lane + 32 * q] = acc[a][q];
  }
}

// dw[tc][ci] (ci < c_in) = sum over the blocks' partials in a fixed order: 64 outputs per workgroup, four threads per output each summing
// every fourth partial (two accumulators), combined through LDS — 400 sequential loads per thread made this a 30-us latency chain
__global__ void __launch_bounds__(256) narrow_wgrad_reduce(const float* __restrict__ part, int n_part, int ntc, int ci_p, int c_in, float* __restrict__ dw) {
  __shared__ float red[256];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  const int64_t stride = (int64_t)ntc * ci_p;
  float s0 = 0.f, s1 = 0.f;
  if (e < ntc * ci_p) {
    int k = kg;
    for (; k + 4 < n_part; k += 8) {
      s0 += part[(int64_t)k * stride + e];
      s1 += part[(int64_t)(k + 4) * stride + e];
    }
    if (k < n_part) s0 += part[(int64_t)k * stride + e];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (kg == 0 && e < ntc * ci_p) {
    const int ci = e % ci_p, tc = e / ci_p;
    if (ci < c_in) dw[(int64_t)tc * c_in + ci] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
  }
}

bool shape_ok(int n, int h, int w, int c_out, int ci_p) {
  return n > 0 && h > 0 && w > 0 && c_out >= 1 && c_out <= 4 && ci_p >= 32 && ci_p <= 32 * MAXQ && ci_p % 32 == 0 && w % 16 == 0 && h % RB == 0 &&
         (int64_t)(TAPS * c_out * ci_p + 7 * 35 * c_out) * 4 <= 64 * 1024 && (int64_t)(w * ci_p + 5 * (2 * w + 3) * c_out) * 4 <= 64 * 1024;
}

}  // namespace

extern "C" int tg_deconv5x5s2_narrow_supported(int n, int h, int w, int c_out, int ci_p) { return shape_ok(n, h, w, c_out, ci_p) ? 1 : 0; }

extern "C" int64_t tg_deconv5x5s2_narrow_wgrad_workspace_bytes(int n, int h, int w, int c_out, int ci_p) {
  if (!shape_ok(n, h, w, c_out, ci_p)) { tg::set_error("deconv5x5s2_narrow: shape not supported"); return TG_ERR_INVALID; }
  return (int64_t)n * (h / RB) * TAPS * c_out * ci_p * 4;
}

extern "C" int tg_deconv5x5s2_narrow_dgrad_f32(const float* dy, int ld_dy, const float* kernel, const float* scale_a, int n, int h, int w, int c_out, int c_in,
                                               int ci_p, float* dx, int ld_dx, void* stream) {
  TG_REQUIRE(dy && kernel && dx, "deconv5x5s2_narrow_dgrad: null buffer");
  TG_REQUIRE(shape_ok(n, h, w, c_out, ci_p) && ld_dy >= c_out && c_in >= 1 && c_in <= ci_p && ld_dx >= ci_p,
             "deconv5x5s2_narrow_dgrad: unsupported shape n=%d h=%d w=%d c_out=%d c_in=%d ci_p=%d", n, h, w, c_out, c_in, ci_p);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_IGEMM, 2.0 * n * h * w * ci_p * TAPS * c_out, 4.0 * ((double)n * h * w * ci_p + (double)n * 4 * h * w * c_out), s, "narrow deconv dgrad");
  const dim3 grid(n * (h / 2) * (w / 16));
  const size_t sh = (size_t)(TAPS * c_out * ci_p + 7 * 35 * c_out) * 4;
  switch (c_out) {
    case 1: hipLaunchKernelGGL(narrow_dgrad<1>, grid, dim3(256), sh, s, dy, ld_dy, kernel, scale_a, c_in, h, w, ci_p, dx, ld_dx); break;
    case 2: hipLaunchKernelGGL(narrow_dgrad<2>, grid, dim3(256), sh, s, dy, ld_dy, kernel, scale_a, c_in, h, w, ci_p, dx, ld_dx); break;
    case 3: hipLaunchKernelGGL(narrow_dgrad<3>, grid, dim3(256), sh, s, dy, ld_dy, kernel, scale_a, c_in, h, w, ci_p, dx, ld_dx); break;
    default: hipLaunchKernelGGL(narrow_dgrad<4>, grid, dim3(256), sh, s, dy, ld_dy, kernel, scale_a, c_in, h, w, ci_p, dx, ld_dx); break;
  }
  TG_CHECK_LAUNCH("narrow_dgrad");
  return TG_OK;
}

extern "C" int tg_deconv5x5s2_narrow_wgrad_f32(const float* dy, int ld_dy, const float* x, int ld_x, int n, int h, int w, int c_out, int c_in, int ci_p,
                                               float* workspace, float* dw, void* stream) {
  TG_REQUIRE(dy && x && workspace && dw, "deconv5x5s2_narrow_wgrad: null buffer");
  TG_REQUIRE(shape_ok(n, h, w, c_out, ci_p) && ld_dy >= c_out && ld_x >= ci_p && c_in >= 1 && c_in <= ci_p, "deconv5x5s2_narrow_wgrad: unsupported shape n=%d h=%d w=%d c_out=%d c_in=%d ci_p=%d",
             n, h, w, c_out, c_in, ci_p);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_WGRAD, 2.0 * n * h * w * ci_p * TAPS * c_out, 4.0 * ((double)n * h * w * ci_p + (double)n * 4 * h * w * c_out), s, "narrow deconv wgrad");
  const int blocks = n * (h / RB);
  const size_t sh = (size_t)(w * ci_p + 5 * (2 * w + 3) * c_out) * 4;
  switch (c_out) {
    case 1: hipLaunchKernelGGL(narrow_wgrad<1>, dim3(blocks), dim3(256), sh, s, dy, ld_dy, x, ld_x, h, w, ci_p, workspace); break;
    case 2: hipLaunchKernelGGL(narrow_wgrad<2>, dim3(blocks), dim3(256), sh, s, dy, ld_dy, x, ld_x, h, w, ci_p, workspace); break;
    case 3: hipLaunchKernelGGL(narrow_wgrad<3>, dim3(blocks), dim3(256), sh, s, dy, ld_dy, x, ld_x, h, w, ci_p, workspace); break;
    default: hipLaunchKernelGGL(narrow_wgrad<4>, dim3(blocks), dim3(256), sh, s, dy, ld_dy, x, ld_x, h, w, ci_p, workspace); break;
  }
  TG_CHECK_LAUNCH("narrow_wgrad");
  const int ntc = TAPS * c_out;
  hipLaunchKernelGGL(narrow_wgrad_reduce, dim3((ntc * ci_p + 63) / 64), dim3(256), 0, s, workspace, blocks, ntc, ci_p, c_in, dw);
  TG_CHECK_LAUNCH("narrow_wgrad_reduce");
  return TG_OK;
}
