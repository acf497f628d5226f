// Backward pass of a 5x5 / stride-2 / 'same' transposed convolution with a NARROW output (the generator's image layer: 138 -> 3 channels,
// Model/Good_GAN_cifar10.py:55-57, Model/Good_GAN.py:76-83; tf.layers.conv2d_transpose, Model/modle_base.py:246-259):
//   dx[n,i,j,ci]     = sum_{ky,kx,co} dy[n, 2i+ky-1, 2j+kx-1, co] * W[ky,kx,co,ci]          (tg_deconv5x5s2_narrow_dgrad_f32)
//   dW[ky,kx,co,ci]  = sum_{n,i,j}    dy[n, 2i+ky-1, 2j+kx-1, co] * x[n,i,j,ci]            (tg_deconv5x5s2_narrow_wgrad_f32)
// (out-of-image dy positions contribute nothing; 'same' padding of the forward op: out[2i+ky-1, 2j+kx-1] += x[i,j] W[ky,kx]).
//
// Why not the generic MFMA kernels: their tiles pad the 3 image channels to 32, so the input gradient (K = 25 taps x 32 "channels") and the
// filter gradient (32 x 160 outputs per tap) do ten times the arithmetic the layer has (round 2: 0.094 + 0.138 ms per step for 0.27 + 0.61
// GFLOP of real work).  Round 3 ran the layer's own products as scalar FMAs (81 + 64 + 17 us); since round 4 they are matrix products again,
// but K-PACKED: the contraction index is the (tap, channel) pair itself (see below).
#include "tg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TAPS = 25;
constexpr int MAXQ = 8;              // 32-channel column tiles of the input channels: ci_p <= 256
constexpr int PC = 35;               // dy patch columns of a 16-pixel block: 2*j0-1 .. 2*j0+33
constexpr int DROWS = 8;             // input-gradient block: up to 8 image rows x 16 pixels = four 32-pixel row tiles, one per wave
constexpr int RB = 4;                // filter-gradient block: image rows per block (one partial slab per block)

__host__ __device__ constexpr int kpad(int co) { return (TAPS * co + 1) & ~1; }      // (tap, co) pairs rounded up to the MFMA's k-pair

// Round 4: both kernels are K-PACKED matrix products on v_mfma_f32_32x32x2_f32 — the contraction index is the (tap, channel) PAIR (75 for the
// three image channels), not taps x 32 padded channels, so the matrix pipe does the layer's own arithmetic (0.61 GFLOP per 100 images) instead
// of ten times it.  The dy patch of a pixel block lies in LDS once; an operand element (pixel, k) is the patch at pixel offset + ktab[k].
// Before (round 3, the same products as scalar FMAs: one ds_read per 2 - 3 FMAs): 81 us + 64 us + 17 us per step alone on the chip.

// ---- input gradient: dx[pixel][ci] = sum_k P[pixel][k] * W[k][ci];  block = up to 8 image rows x 16 pixels, wave = one 32-pixel row pair ----
template <int CO>
__global__ void __launch_bounds__(256) narrow_dgrad(const float* __restrict__ dy, int ld_dy, const float* __restrict__ kernel, const float* __restrict__ scale_a,
                                                    int c_in, int h, int w, int ci_p, float* __restrict__ dx, int ld_dx) {
  constexpr int K = TAPS * CO, KP = kpad(CO), PR = 2 * DROWS + 3;
  extern __shared__ float lds[];
  float* Wl = lds;                                   // [KP][ci_p]: the variable's own [5,5,Cout,Cin] rows (x the weight-norm scale), zero beyond K and c_in
  float* Pl = lds + KP * ci_p;                       // [PR][PC][CO] dy patch: rows 2*i0-1 .. 2*i0+2*DROWS+1, columns 2*j0-1 .. 2*j0+33
  int* ktab = reinterpret_cast<int*>(Pl + PR * PC * CO);     // [KP] patch offset of (tap, co) pair k
  const int tiles_x = w / 16, tiles_y = (h + DROWS - 1) / DROWS;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int i0 = ty * DROWS, j0 = tx * 16;
  const int tid = threadIdx.x, lane = tid & 31, pg = tid >> 5;
  const int nq = ci_p / 32;
  // (staging loads go out in groups — two filter rows x nq units, eight patch elements — before anything is stored: one load per loop trip made
  // the prologue a chain of memory latencies)
  for (int tc0 = pg; tc0 < KP; tc0 += 16) {          // half-wave = one filter row: coalesced 128-byte reads
    float v[2][MAXQ];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tc = tc0 + 8 * u;
      const float sc = (scale_a && tc < K) ? scale_a[tc % CO] : 1.f;
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int ci = lane + 32 * q;
        v[u][q] = (q < nq && tc < K && ci < c_in) ? kernel[(int64_t)tc * c_in + ci] * sc : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tc = tc0 + 8 * u;
#pragma unroll
      for (int q = 0; q < MAXQ; ++q)
        if (q < nq && tc < KP) Wl[tc * ci_p + lane + 32 * q] = v[u][q];
    }
  }
  const int H2 = 2 * h, W2 = 2 * w;
  for (int e0 = tid; e0 < PR * PC * CO; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      const int co = e % CO, c = (e / CO) % PC, r = e / (CO * PC);
      const int oy = 2 * i0 - 1 + r, ox = 2 * j0 - 1 + c;
      v[u] = (e < PR * PC * CO && (unsigned)oy < (unsigned)H2 && (unsigned)ox < (unsigned)W2) ? dy[(((int64_t)n * H2 + oy) * W2 + ox) * ld_dy + co] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e < PR * PC * CO) Pl[e] = v[u];
    }
  }
  if (tid < KP) {
    const int tap = tid / CO, co = tid - tap * CO, ky = tap / 5, kx = tap - ky * 5;
    ktab[tid] = tid < K ? (ky * PC + kx) * CO + co : 0;          // k >= K multiplies a zero filter row: any patch element will do
  }
  __syncthreads();
  const int wv = tid >> 6, l = tid & 63, m = l & 31, hl = l >> 5;
  const int rows_blk = min(DROWS, h - i0);                       // even: h is
  if (2 * wv >= rows_blk) return;
  // A[pixel m][k]: pixel row 2*wv + (m >> 4), column m & 15 of the block -> patch element (2*row + ky, 2*col + kx)
  const float* pa = Pl + ((2 * (2 * wv + (m >> 4))) * PC + 2 * (m & 15)) * CO;
  const float* pb = Wl + hl * ci_p + m;                          // B[k = 2s + hl][ci = 32q + m]
  f32x16 acc[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  int ko = ktab[hl];
  for (int s = 0; s < KP / 2; ++s) {
    const float a = pa[ko];
    if (s + 1 < KP / 2) ko = ktab[2 * (s + 1) + hl];
    const float* wb = pb + 2 * s * ci_p;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
      if (q < nq) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[32 * q], acc[q], 0, 0, 0);      // D[pixel][ci]: lane = channel
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mo = (r & 3) + 8 * (r >> 2) + 4 * hl;              // pixel of accumulator register r
    const int i = i0 + 2 * wv + (mo >> 4), j = j0 + (mo & 15);
    float* o = dx + (((int64_t)n * h + i) * w + j) * ld_dx + m;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
      if (q < nq) o[32 * q] = acc[q][r];
  }
}

// ---- filter gradient: dW[k][ci] = sum_pixels P[pixel][k] * x[pixel][ci];  block = RB image rows of one image, the (k tile, channel tile) pairs
// dealt to the four waves; one image row in LDS at a time, the next one on its way in registers -------------------------------------------
template <int CO, int XU>
__global__ void __launch_bounds__(256, 2) narrow_wgrad(const float* __restrict__ dy, int ld_dy, const float* __restrict__ x, int ld_x, int h, int w, int ci_p,
                                                    float* __restrict__ part) {
  constexpr int K = TAPS * CO, KP = kpad(CO), NMT = (KP + 31) / 32, PERW = (NMT * MAXQ + 3) / 4;
  extern __shared__ float lds[];
  float* Xl = lds;                                     // [w][ci_p] one image row of x
  float* Pl = lds + w * ci_p;                          // [5][2w + 3][CO] the dy rows 2i-1 .. 2i+3
  const int tiles_y = h / RB;
  const int ty = blockIdx.x % tiles_y, n = blockIdx.x / tiles_y;
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, m = l & 31, hl = l >> 5;
  const int nq = ci_p / 32, PW = 2 * w + 3, H2 = 2 * h, W2 = 2 * w;
  const int n_tiles = NMT * nq;
  // tile a of this wave: (k tile mt, channel tile nt); A[k = 32 mt + m][pixel j] = patch[koff(k) + 2 j CO], B[pixel j][ci = 32 nt + m]
  int a_off[PERW], b_off[PERW];
#pragma unroll
  for (int a = 0; a < PERW; ++a) {
    const int t = wv + 4 * a, mt = t / nq, nt = t - mt * nq;
    const int k = 32 * mt + m;
    const int tap = k / CO, co = k - tap * CO, ky = tap / 5, kx = tap - ky * 5;
    a_off[a] = (k < K ? (ky * PW + kx) * CO + co : 0) + hl * 2 * CO;        // rows k >= K are computed on some in-range element and never stored
    b_off[a] = hl * ci_p + 32 * nt + m;
  }
  f32x16 acc[PERW];
#pragma unroll
  for (int a = 0; a < PERW; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  // staging: x row as 16-byte units (ci_p / 4 per pixel), the patch element by element; XU / PU units per thread
  const int xunits = w * (ci_p / 4), punits = 5 * PW * CO;
  constexpr int PU = 6;                                // XU x 256 16-byte units of an x row (w * ci_p / 4 <= 2048: XU = 8; <= 1024: XU = 4); 5 * 67 * 4 = 1340 patch elements / 256
  f32x4 xr[XU];
  float pr[PU];
  auto gload = [&](int i) {
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = tid + 256 * u < xunits ? tid + 256 * u : 0;   // (units beyond the row re-read unit 0 and are not stored: keeps xr in registers)
      const int j = e / (ci_p / 4), c4 = e - j * (ci_p / 4);
      xr[u] = *reinterpret_cast<const f32x4*>(x + (((int64_t)n * h + i) * w + j) * ld_x + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int e = tid + 256 * u;
      const int co = e % CO, c = (e / CO) % PW, r = e / (CO * PW);
      const int oy = 2 * i - 1 + r, ox = c - 1;
      pr[u] = (e < punits && (unsigned)oy < (unsigned)H2 && (unsigned)ox < (unsigned)W2) ? dy[(((int64_t)n * H2 + oy) * W2 + ox) * ld_dy + co] : 0.f;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = tid + 256 * u;
      if (e < xunits) *reinterpret_cast<f32x4*>(Xl + 4 * e) = xr[u];
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int e = tid + 256 * u;
      if (e < punits) Pl[e] = pr[u];
    }
  };
  const int i_first = ty * RB;
  gload(i_first);
  for (int i = i_first; i < i_first + RB; ++i) {
    __syncthreads();                                   // the previous row's fragments have been read
    sstore();
    __syncthreads();
    if (i + 1 < i_first + RB) gload(i + 1);            // in flight while this row is multiplied
    for (int s = 0; s < w / 2; ++s) {                  // pixel pair (2s, 2s + 1): lane half hl takes pixel 2s + hl
#pragma unroll
      for (int a = 0; a < PERW; ++a)
        if (wv + 4 * a < n_tiles)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(Pl[a_off[a] + 4 * s * CO], Xl[b_off[a] + 2 * s * ci_p], acc[a], 0, 0, 0);   // D[k][ci]
    }
  }
  float* o = part + (int64_t)blockIdx.x * K * ci_p;
#pragma unroll
  for (int a = 0; a < PERW; ++a) {
    const int t = wv + 4 * a;
    if (t < n_tiles) {
      const int mt = t / nq, nt = t - mt * nq;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tc = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hl;
        if (tc < K) o[tc * ci_p + 32 * nt + m] = acc[a][r];
      }
    }
  }
}

// dw[tc][ci] (ci < c_in) = sum over the blocks' partials in a fixed order: 64 outputs per workgroup, sixteen threads per output each summing
// every sixteenth partial (two accumulators), combined through LDS in a fixed tree
__global__ void __launch_bounds__(1024) narrow_wgrad_reduce(const float* __restrict__ part, int n_part, int ntc, int ci_p, int c_in, float* __restrict__ dw) {
  __shared__ float red[1024];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), kg = threadIdx.x >> 6;
  const int64_t stride = (int64_t)ntc * ci_p;
  float s0 = 0.f, s1 = 0.f;
  if (e < ntc * ci_p) {
    int k = kg;
    for (; k + 16 < n_part; k += 32) {
      s0 += part[(int64_t)k * stride + e];
      s1 += part[(int64_t)(k + 16) * stride + e];
    }
    if (k < n_part) s0 += part[(int64_t)k * stride + e];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (kg < 4) red[threadIdx.x] = (red[threadIdx.x] + red[threadIdx.x + 256]) + (red[threadIdx.x + 512] + red[threadIdx.x + 768]);
  __syncthreads();
  if (kg == 0 && e < ntc * ci_p) {
    const int ci = e % ci_p, tc = e / ci_p;
    if (ci < c_in) dw[(int64_t)tc * c_in + ci] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
  }
}

size_t dgrad_lds_bytes(int c_out, int ci_p) { return (size_t)(kpad(c_out) * ci_p + (2 * DROWS + 3) * PC * c_out + kpad(c_out)) * 4; }
size_t wgrad_lds_bytes(int w, int c_out, int ci_p) { return (size_t)(w * ci_p + 5 * (2 * w + 3) * c_out) * 4; }

bool shape_ok(int n, int h, int w, int c_out, int ci_p) {
  return n > 0 && h > 0 && w > 0 && c_out >= 1 && c_out <= 4 && ci_p >= 32 && ci_p <= 32 * MAXQ && ci_p % 32 == 0 && w % 16 == 0 && w <= 32 && h % RB == 0 &&
         dgrad_lds_bytes(c_out, ci_p) <= 64 * 1024 && wgrad_lds_bytes(w, c_out, ci_p) <= 64 * 1024;
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
  const dim3 grid(n * ((h + DROWS - 1) / DROWS) * (w / 16));
  const size_t sh = dgrad_lds_bytes(c_out, ci_p);
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
  const size_t sh = wgrad_lds_bytes(w, c_out, ci_p);
  const bool big = w * (ci_p / 4) > 1024;              // staging registers per thread: 4 or 8 16-byte units of an x row
#define TG_NARROW_WGRAD(CO_)                                                                                                            \
  if (big) hipLaunchKernelGGL((narrow_wgrad<CO_, 8>), dim3(blocks), dim3(256), sh, s, dy, ld_dy, x, ld_x, h, w, ci_p, workspace);       \
  else hipLaunchKernelGGL((narrow_wgrad<CO_, 4>), dim3(blocks), dim3(256), sh, s, dy, ld_dy, x, ld_x, h, w, ci_p, workspace)
  switch (c_out) {
    case 1: TG_NARROW_WGRAD(1); break;
    case 2: TG_NARROW_WGRAD(2); break;
    case 3: TG_NARROW_WGRAD(3); break;
    default: TG_NARROW_WGRAD(4); break;
  }
#undef TG_NARROW_WGRAD
  TG_CHECK_LAUNCH("narrow_wgrad");
  const int ntc = TAPS * c_out;
  hipLaunchKernelGGL(narrow_wgrad_reduce, dim3((ntc * ci_p + 63) / 64), dim3(1024), 0, s, workspace, blocks, ntc, ci_p, c_in, dw);
  TG_CHECK_LAUNCH("narrow_wgrad_reduce");
  return TG_OK;
}
