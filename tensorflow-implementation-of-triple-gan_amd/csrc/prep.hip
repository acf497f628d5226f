// Parameter-side kernels: weight-norm reparameterisation (forward + backward), filter re-layout with
// channel padding for the MFMA kernels, and the deterministic split-K slab reduction of tg_wgrad_f32.
// Column (= output channel) reductions keep 32 consecutive channels per wave row for 128-B coalesced
// reads and finish with an LDS tree; the re-layout is a padded 32x32 LDS tile transpose.
#include "tg_common.h"
#include "tg_device.h"

namespace {

// scale[c] = g[c] * rsqrt(max(sum_r V[r][c]^2, 1e-12))        (tf.nn.l2_normalize, Model/nn.py:502)
// 32 columns x 32 row-lanes per 1024-thread block: the row loop is a dependent-load chain, so it is kept short
// (rows/32 steps, 4 loads in flight per lane) — a filter has only c/32 = 4..16 such column groups.
__global__ void __launch_bounds__(1024) wn_scale(const float* __restrict__ v, const float* __restrict__ g, int r, int c, float* __restrict__ scale) {
  const int tx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < c) {
    int i = ry;
    for (; i + 96 < r; i += 128) {
      const float t0 = v[(int64_t)i * c + col], t1 = v[(int64_t)(i + 32) * c + col], t2 = v[(int64_t)(i + 64) * c + col],
                  t3 = v[(int64_t)(i + 96) * c + col];
      a0 += t0 * t0; a1 += t1 * t1; a2 += t2 * t2; a3 += t3 * t3;
    }
    for (; i < r; i += 32) { const float t = v[(int64_t)i * c + col]; a0 += t * t; }
  }
  float acc = (a0 + a1) + (a2 + a3);
  __shared__ float red[32][33];
  red[ry][tx] = acc;
  __syncthreads();
  if (ry == 0 && col < c) {
    for (int k = 1; k < 32; ++k) acc += red[k][tx];
    scale[col] = g[col] * rsqrtf(fmaxf(acc, 1e-12f));
  }
}

// dg[c] = <dW[:,c], V[:,c]> / ||V[:,c]|| ;  coef[c] = {g/||V||, <dW,V>/||V||^2}
__global__ void __launch_bounds__(1024) wn_bwd_cols(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g, int r, int c,
                                                    float* __restrict__ dg, float* __restrict__ coef) {
  const int tx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  float d0 = 0.f, d1 = 0.f, s0 = 0.f, s1 = 0.f;
  if (col < c) {
    int i = ry;
    for (; i + 32 < r; i += 64) {
      const float t0 = v[(int64_t)i * c + col], t1 = v[(int64_t)(i + 32) * c + col];
      const float w0 = dw[(int64_t)i * c + col], w1 = dw[(int64_t)(i + 32) * c + col];
      d0 += w0 * t0; d1 += w1 * t1; s0 += t0 * t0; s1 += t1 * t1;
    }
    for (; i < r; i += 32) {
      const float t = v[(int64_t)i * c + col];
      d0 += dw[(int64_t)i * c + col] * t;
      s0 += t * t;
    }
  }
  float dot = d0 + d1, ss = s0 + s1;
  __shared__ float red[2][32][33];
  red[0][ry][tx] = dot;
  red[1][ry][tx] = ss;
  __syncthreads();
  if (ry == 0 && col < c) {
    for (int k = 1; k < 32; ++k) { dot += red[0][k][tx]; ss += red[1][k][tx]; }
    const float nrm = sqrtf(ss);
    dg[col] = dot / nrm;
    coef[col] = g[col] / nrm;
    coef[c + col] = dot / ss;
  }
}

__global__ void __launch_bounds__(256) wn_bwd_apply(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ coef, int64_t n,
                                                    int c, float* __restrict__ dv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % c);
    dv[i] = coef[col] * (dw[i] - v[i] * coef[c + col]);
  }
}

constexpr int MAX_PREP_JOBS = 24;
struct PrepJobs { tg_prep_job j[MAX_PREP_JOBS]; int first_block[MAX_PREP_JOBS + 1]; int n; };

// scale[c] = g[c] / ||V[:,c]||  (rows = t*a), blockIdx.y = job
__global__ void __launch_bounds__(1024) wn_scale_multi(PrepJobs js) {
  const tg_prep_job& J = js.j[blockIdx.y];
  if (J.g == nullptr) return;
  const int r = J.t * J.a, c = J.b;
  if (blockIdx.x * 32 >= c) return;
  const int tx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                 // same summation order as wn_scale: both paths give identical bits
  const float* v = J.src;
  if (col < c) {
    int i = ry;
    for (; i + 96 < r; i += 128) {
      const float t0 = v[(int64_t)i * c + col], t1 = v[(int64_t)(i + 32) * c + col], t2 = v[(int64_t)(i + 64) * c + col],
                  t3 = v[(int64_t)(i + 96) * c + col];
      a0 += t0 * t0; a1 += t1 * t1; a2 += t2 * t2; a3 += t3 * t3;
    }
    for (; i < r; i += 32) { const float t = v[(int64_t)i * c + col]; a0 += t * t; }
  }
  float acc = (a0 + a1) + (a2 + a3);
  __shared__ float red[32][33];
  red[ry][tx] = acc;
  __syncthreads();
  if (ry == 0 && col < c) {
    for (int k = 1; k < 32; ++k) acc += red[k][tx];
    J.scale[col] = J.g[col] * rsqrtf(fmaxf(acc, 1e-12f));
  }
}

// filter_prep for every job; workgroup -> (job, t, a-tile, b-tile) through the first_block prefix table
__global__ void __launch_bounds__(256) filter_prep_multi(PrepJobs js) {
  __shared__ float tile[32][33];
  int job = 0;
  while (job + 1 < js.n && (int)blockIdx.x >= js.first_block[job + 1]) ++job;
  const tg_prep_job& J = js.j[job];
  int rem = blockIdx.x - js.first_block[job];
  const int at = (J.a_pad + 31) / 32, bt = (J.b_pad + 31) / 32;
  const int ai = rem % at; rem /= at;
  const int bi = rem % bt;
  const int t = rem / bt;
  const int a0 = ai * 32, b0 = bi * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* scale = J.g ? J.scale : nullptr;
  for (int i = ty; i < 32; i += 8) {
    const int a = a0 + i, b = b0 + tx;
    float v = 0.f;
    if (a < J.a && b < J.b) {
      v = J.src[((int64_t)t * J.a + a) * J.b + b];
      if (scale) v *= scale[b];
    }
    tile[i][tx] = v;
    if (J.dst_same && a < J.a_pad && b < J.b_pad) J.dst_same[((int64_t)t * J.a_pad + a) * J.b_pad + b] = v;
  }
  __syncthreads();
  if (J.dst_tr) {
    for (int i = ty; i < 32; i += 8) {
      const int b = b0 + i, a = a0 + tx;
      if (b < J.b_pad && a < J.a_pad) J.dst_tr[(int64_t)b * J.tr_sb + (int64_t)t * J.tr_st + a] = tile[tx][i];
    }
  }
}

constexpr int MAX_WN_JOBS = 16;
struct WnJobs { tg_wn_job j[MAX_WN_JOBS]; };

// blockIdx.y = job
__global__ void __launch_bounds__(256) slab_reduce_multi(WnJobs js) {
  const tg_wn_job& J = js.j[blockIdx.y];
  const int64_t total = (int64_t)J.t * J.c_in * J.c_out;
  const int64_t sstride = (int64_t)J.t * J.c_pad * J.n_pad;
  if ((J.c_out & 3) == 0 && (J.n_pad & 3) == 0 && ((reinterpret_cast<uintptr_t>(J.slab) | reinterpret_cast<uintptr_t>(J.dw)) & 15) == 0) {
    // four consecutive columns per thread: 16-byte loads (a quarter of the requests for the same bytes: the classifier's slabs are 190 MB
    // per step); every output is still the sum of its n_split partials in slab order — bit-identical to the scalar form below
    const int64_t total4 = total >> 2;
    const int q = J.c_out >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
      const int n4 = (int)(i % q);
      const int64_t tc = i / q;
      const int c = (int)(tc % J.c_in), t = (int)(tc / J.c_in);
      const float* p = J.slab + ((int64_t)t * J.c_pad + c) * J.n_pad + 4 * n4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < J.n_split; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(p + s * sstride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(J.dw + 4 * i) = acc;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % J.c_out);
    const int64_t tc = i / J.c_out;
    const int c = (int)(tc % J.c_in), t = (int)(tc / J.c_in);
    const float* p = J.slab + ((int64_t)t * J.c_pad + c) * J.n_pad + n;
    float acc = 0.f;
    for (int s = 0; s < J.n_split; ++s) acc += p[s * sstride];
    J.dw[i] = acc;
  }
}

__global__ void __launch_bounds__(1024) wn_bwd_cols_multi(WnJobs js) {
  const tg_wn_job& J = js.j[blockIdx.y];
  if (J.v == nullptr) return;
  const int r = J.t * J.c_in, c = J.c_out;
  if (blockIdx.x * 32 >= c) return;
  const int tx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  float d0 = 0.f, s0 = 0.f;
  if (col < c) {
    for (int i = ry; i < r; i += 32) {
      const float t = J.v[(int64_t)i * c + col];
      d0 += J.dw[(int64_t)i * c + col] * t;
      s0 += t * t;
    }
  }
  __shared__ float red[2][32][33];
  red[0][ry][tx] = d0;
  red[1][ry][tx] = s0;
  __syncthreads();
  if (ry == 0 && col < c) {
    float dot = d0, ss = s0;
    for (int k = 1; k < 32; ++k) { dot += red[0][k][tx]; ss += red[1][k][tx]; }
    const float nrm = sqrtf(ss);
    J.dg[col] = dot / nrm;
    J.coef[col] = J.g[col] / nrm;
    J.coef[c + col] = dot / ss;
  }
}

__global__ void __launch_bounds__(256) wn_bwd_apply_multi(WnJobs js) {
  const tg_wn_job& J = js.j[blockIdx.y];
  if (J.v == nullptr) return;
  const int c = J.c_out;
  const int64_t n = (int64_t)J.t * J.c_in * c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % c);
    J.dv[i] = J.coef[col] * (J.dw[i] - J.v[i] * J.coef[c + col]);
  }
}

// src [T][A][B] (B contiguous), optional per-b scale.
//   dst_same[t][a][b]      padded copy  [T][A_pad][B_pad]
//   dst_tr  [b*sb + t*st + a]           (a contiguous; rows b < B_pad, a < A_pad, zero padded)
__global__ void __launch_bounds__(256) filter_prep(const float* __restrict__ src, const float* __restrict__ scale, const float* __restrict__ scale_a,
                                                   int a_dim, int b_dim, int a_pad, int b_pad, float* __restrict__ dst_same, float* __restrict__ dst_tr,
                                                   int64_t sb, int64_t st) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int a = a0 + i, b = b0 + tx;
    float v = 0.f;
    if (a < a_dim && b < b_dim) {
      v = src[((int64_t)t * a_dim + a) * b_dim + b];
      if (scale) v *= scale[b];
      if (scale_a) v *= scale_a[a];
    }
    tile[i][tx] = v;
    if (dst_same && a < a_pad && b < b_pad) dst_same[((int64_t)t * a_pad + a) * b_pad + b] = v;
  }
  __syncthreads();
  if (dst_tr) {
    for (int i = ty; i < 32; i += 8) {
      const int b = b0 + i, a = a0 + tx;
      if (b < b_pad && a < a_pad) dst_tr[(int64_t)b * sb + (int64_t)t * st + a] = tile[tx][i];
    }
  }
}

struct TapMap { int32_t m[36]; };

// merged transposed-conv filter (see tg_deconv_merge_prep_f32): one thread per destination element, c fastest
__global__ void __launch_bounds__(256) deconv_merge_prep(const float* __restrict__ w, const float* __restrict__ scale_a, int c_out, int c_in,
                                                         int n_group, int n_pad, int c_pad, TapMap tm, float* __restrict__ dst) {
  const int64_t total = (int64_t)n_pad * 9 * c_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c_pad);
    const int64_t nt = i / c_pad;
    const int t9 = (int)(nt % 9), n = (int)(nt / 9);
    const int g = n / n_group, co = n - g * n_group;
    float v = 0.f;
    if (g < 4 && co < c_out && c < c_in) {
      const int tap = tm.m[g * 9 + t9];
      if (tap >= 0) {
        v = w[((int64_t)tap * c_out + co) * c_in + c];
        if (scale_a) v *= scale_a[co];
      }
    }
    dst[i] = v;
  }
}

// dst[t][c][n] = sum_s slab[s][t][c][n], c < C, n < N  (fixed summation order -> bitwise reproducible)
__global__ void __launch_bounds__(256) slab_reduce(const float* __restrict__ slab, int n_split, int t_dim, int c_pad, int n_pad, int c_dim, int n_dim,
                                                   float* __restrict__ dst) {
  const int64_t total = (int64_t)t_dim * c_dim * n_dim;
  const int64_t sstride = (int64_t)t_dim * c_pad * n_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % n_dim);
    const int64_t tc = i / n_dim;
    const int c = (int)(tc % c_dim), t = (int)(tc / c_dim);
    const float* p = slab + ((int64_t)t * c_pad + c) * n_pad + n;
    float acc = 0.f;
    for (int s = 0; s < n_split; ++s) acc += p[s * sstride];
    dst[i] = acc;
  }
}

// Few outputs, many slabs (the 3->128 first convolution: 3 456 outputs x 512 pixel splits): one thread per output would walk the
// slabs serially with 14 workgroups on the chip.  Here 8 lanes share an output, each sums a contiguous range of slabs in order
// and the 8 partial sums are combined in lane order — still one fixed summation order, bitwise reproducible.
__global__ void __launch_bounds__(256) slab_reduce_wide(const float* __restrict__ slab, int n_split, int t_dim, int c_pad, int n_pad, int c_dim,
                                                        int n_dim, float* __restrict__ dst) {
  __shared__ float part[8][32];
  const int64_t total = (int64_t)t_dim * c_dim * n_dim;
  const int64_t sstride = (int64_t)t_dim * c_pad * n_pad;
  const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + o;
  float acc = 0.f;
  if (i < total) {
    const int n = (int)(i % n_dim);
    const int64_t tc = i / n_dim;
    const int c = (int)(tc % c_dim), t = (int)(tc / c_dim);
    const float* p = slab + ((int64_t)t * c_pad + c) * n_pad + n;
    const int per = (n_split + 7) / 8;
    const int s0 = grp * per, s1 = min(n_split, s0 + per);
    for (int s = s0; s < s1; ++s) acc += p[s * sstride];
  }
  part[grp][o] = acc;
  __syncthreads();
  if (grp == 0 && i < total) {
    float r = part[0][o];
#pragma unroll
    for (int g = 1; g < 8; ++g) r += part[g][o];
    dst[i] = r;
  }
}

// weight norm of a transposed-conv filter V[t][a][b] = [kh*kw][Cout][Cin]: the norm runs over axes (0,1,3) = (t, b) for
// every output channel a (Model/modle_base.py:148).  One workgroup per output channel (Cout is 3 in the reference).
__device__ float block_sum256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) wn_scale_tab(const float* __restrict__ v, const float* __restrict__ g, int t_dim, int a_dim, int b_dim,
                                                    float* __restrict__ scale) {
  __shared__ float red[4];
  const int a = blockIdx.x;
  float ss = 0.f;
  for (int i = threadIdx.x; i < t_dim * b_dim; i += 256) {
    const float x = v[((int64_t)(i / b_dim) * a_dim + a) * b_dim + i % b_dim];
    ss += x * x;
  }
  ss = block_sum256(ss, red);
  if (threadIdx.x == 0) scale[a] = g[a] * rsqrtf(fmaxf(ss, 1e-12f));
}

__global__ void __launch_bounds__(256) wn_bwd_tab(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g, int t_dim, int a_dim,
                                                  int b_dim, float* __restrict__ dv, float* __restrict__ dg) {
  __shared__ float red[4];
  const int a = blockIdx.x;
  float dot = 0.f, ss = 0.f;
  for (int i = threadIdx.x; i < t_dim * b_dim; i += 256) {
    const int64_t idx = ((int64_t)(i / b_dim) * a_dim + a) * b_dim + i % b_dim;
    dot += dw[idx] * v[idx];
    ss += v[idx] * v[idx];
  }
  dot = block_sum256(dot, red);
  ss = block_sum256(ss, red);
  const float nrm = sqrtf(ss);
  if (threadIdx.x == 0) dg[a] = dot / nrm;
  const float c0 = g[a] / nrm, c1 = dot / ss;
  for (int i = threadIdx.x; i < t_dim * b_dim; i += 256) {
    const int64_t idx = ((int64_t)(i / b_dim) * a_dim + a) * b_dim + i % b_dim;
    dv[idx] = c0 * (dw[idx] - v[idx] * c1);
  }
}

int ew_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" {

int tg_wn_scale_f32(const float* v, const float* g, int rows, int c, float* scale, void* stream) {
  TG_REQUIRE(v && g && scale && rows > 0 && c > 0, "wn_scale: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * rows * c, s);
  hipLaunchKernelGGL(wn_scale, dim3((c + 31) / 32), dim3(1024), 0, s, v, g, rows, c, scale);
  TG_CHECK_LAUNCH("wn_scale");
  return TG_OK;
}

int tg_filter_prep_f32(const float* src, const float* scale, const float* scale_a, int t, int a, int b, int a_pad, int b_pad, float* dst_same,
                       float* dst_tr, int64_t tr_sb, int64_t tr_st, void* stream) {
  TG_REQUIRE(src && (dst_same || dst_tr) && t > 0 && a > 0 && b > 0 && a_pad >= a && b_pad >= b, "filter_prep: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * t * ((double)a * b + (double)a_pad * b_pad * ((dst_same ? 1 : 0) + (dst_tr ? 1 : 0))), s);
  hipLaunchKernelGGL(filter_prep, dim3((a_pad + 31) / 32, (b_pad + 31) / 32, t), dim3(256), 0, s, src, scale, scale_a, a, b, a_pad, b_pad, dst_same,
                     dst_tr, tr_sb, tr_st);
  TG_CHECK_LAUNCH("filter_prep");
  return TG_OK;
}

int tg_deconv_merge_prep_f32(const float* w, const float* scale_a, int c_out, int c_in, int n_group, int n_pad, int c_pad,
                             const int32_t* tapmap, float* dst, void* stream) {
  TG_REQUIRE(w && dst && tapmap && c_out > 0 && c_in > 0 && n_group >= c_out && n_pad >= 4 * n_group && c_pad >= c_in, "deconv_merge_prep: bad args");
  TapMap tm;
  for (int i = 0; i < 36; ++i) { tm.m[i] = tapmap[i]; TG_REQUIRE(tapmap[i] < 25, "deconv_merge_prep: tap %d out of range", tapmap[i]); }
  hipStream_t s = tg::as_stream(stream);
  const int64_t total = (int64_t)n_pad * 9 * c_pad;
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * (total + 25.0 * c_out * c_in), s);
  hipLaunchKernelGGL(deconv_merge_prep, dim3(ew_grid(total)), dim3(256), 0, s, w, scale_a, c_out, c_in, n_group, n_pad, c_pad, tm, dst);
  TG_CHECK_LAUNCH("deconv_merge_prep");
  return TG_OK;
}

int tg_filter_prep_multi_f32(const tg_prep_job* jobs, int n_jobs, void* stream) {
  TG_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= MAX_PREP_JOBS, "filter_prep_multi: n_jobs=%d out of range", n_jobs);
  PrepJobs js;
  js.n = n_jobs;
  int blocks = 0, max_b = 0;
  bool any_wn = false;
  double bytes = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const tg_prep_job& j = jobs[i];
    TG_REQUIRE(j.src && (j.dst_same || j.dst_tr) && j.t > 0 && j.a > 0 && j.b > 0 && j.a_pad >= j.a && j.b_pad >= j.b, "filter_prep_multi: job %d bad", i);
    TG_REQUIRE(j.g == nullptr || j.scale != nullptr, "filter_prep_multi: job %d is weight-normalised but has no scale buffer", i);
    js.j[i] = j;
    js.first_block[i] = blocks;
    blocks += ((j.a_pad + 31) / 32) * ((j.b_pad + 31) / 32) * j.t;
    max_b = j.b > max_b ? j.b : max_b;
    any_wn = any_wn || j.g != nullptr;
    bytes += 4.0 * j.t * ((double)j.a * j.b * (j.g ? 2 : 1) + (double)j.a_pad * j.b_pad * ((j.dst_same ? 1 : 0) + (j.dst_tr ? 1 : 0)));
  }
  js.first_block[n_jobs] = blocks;
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, bytes, s);
  if (any_wn) {
    hipLaunchKernelGGL(wn_scale_multi, dim3((max_b + 31) / 32, n_jobs), dim3(1024), 0, s, js);
    TG_CHECK_LAUNCH("wn_scale_multi");
  }
  hipLaunchKernelGGL(filter_prep_multi, dim3(blocks), dim3(256), 0, s, js);
  TG_CHECK_LAUNCH("filter_prep_multi");
  return TG_OK;
}

int tg_filter_grad_tail_multi_f32(const tg_wn_job* jobs, int n_jobs, void* stream) {
  TG_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= MAX_WN_JOBS, "filter_grad_tail_multi: n_jobs=%d out of range", n_jobs);
  WnJobs js;
  int64_t max_total = 0;
  int max_c = 0;
  bool any_wn = false;
  double bytes = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const tg_wn_job& j = jobs[i];
    TG_REQUIRE(j.slab && j.dw && j.n_split >= 1 && j.t >= 1 && j.c_in >= 1 && j.c_out >= 1 && j.c_in <= j.c_pad && j.c_out <= j.n_pad,
               "filter_grad_tail_multi: job %d has bad geometry", i);
    TG_REQUIRE(j.v == nullptr || (j.g && j.dv && j.dg && j.coef), "filter_grad_tail_multi: job %d lacks weight-norm buffers", i);
    js.j[i] = j;
    const int64_t total = (int64_t)j.t * j.c_in * j.c_out;
    max_total = total > max_total ? total : max_total;
    max_c = j.c_out > max_c ? j.c_out : max_c;
    any_wn = any_wn || j.v != nullptr;
    bytes += 4.0 * total * (j.n_split + 1 + (j.v ? 5 : 0));
  }
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, bytes, s);
  int gx = (int)((max_total + 255) / 256);
  gx = gx > 1024 ? 1024 : gx;
  hipLaunchKernelGGL(slab_reduce_multi, dim3(gx, n_jobs), dim3(256), 0, s, js);
  TG_CHECK_LAUNCH("slab_reduce_multi");
  if (any_wn) {
    hipLaunchKernelGGL(wn_bwd_cols_multi, dim3((max_c + 31) / 32, n_jobs), dim3(1024), 0, s, js);
    TG_CHECK_LAUNCH("wn_bwd_cols_multi");
    hipLaunchKernelGGL(wn_bwd_apply_multi, dim3(gx, n_jobs), dim3(256), 0, s, js);
    TG_CHECK_LAUNCH("wn_bwd_apply_multi");
  }
  return TG_OK;
}

int tg_slab_reduce_f32(const float* slab, int n_split, int t, int c_pad, int n_pad, int c, int n, float* dst, void* stream) {
  TG_REQUIRE(slab && dst && n_split >= 1 && c <= c_pad && n <= n_pad, "slab_reduce: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * t * c * n * (n_split + 1), s);
  const int64_t total = (int64_t)t * c * n;
  if (n_split >= 32 && total <= 65536)
    hipLaunchKernelGGL(slab_reduce_wide, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, slab, n_split, t, c_pad, n_pad, c, n, dst);
  else
    hipLaunchKernelGGL(slab_reduce, dim3(ew_grid(total)), dim3(256), 0, s, slab, n_split, t, c_pad, n_pad, c, n, dst);
  TG_CHECK_LAUNCH("slab_reduce");
  return TG_OK;
}

int tg_wn_scale_tab_f32(const float* v, const float* g, int t, int a, int b, float* scale_a, void* stream) {
  TG_REQUIRE(v && g && scale_a && t > 0 && a > 0 && b > 0, "wn_scale_tab: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * t * a * b, s);
  hipLaunchKernelGGL(wn_scale_tab, dim3(a), dim3(256), 0, s, v, g, t, a, b, scale_a);
  TG_CHECK_LAUNCH("wn_scale_tab");
  return TG_OK;
}

int tg_wn_bwd_tab_f32(const float* dw, const float* v, const float* g, int t, int a, int b, float* dv, float* dg, void* stream) {
  TG_REQUIRE(dw && v && g && dv && dg && t > 0 && a > 0 && b > 0, "wn_bwd_tab: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * t * a * b * 5, s);
  hipLaunchKernelGGL(wn_bwd_tab, dim3(a), dim3(256), 0, s, dw, v, g, t, a, b, dv, dg);
  TG_CHECK_LAUNCH("wn_bwd_tab");
  return TG_OK;
}

/* coef: scratch of 2*c floats */
int tg_wn_bwd_f32(const float* dw, const float* v, const float* g, int rows, int c, float* dv, float* dg, float* coef, void* stream) {
  TG_REQUIRE(dw && v && g && dv && dg && coef, "wn_bwd: null buffer");
  TG_REQUIRE(rows > 0 && c > 0, "wn_bwd: rows=%d c=%d", rows, c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_PREP, 0, 4.0 * rows * c * 5, s);
  hipLaunchKernelGGL(wn_bwd_cols, dim3((c + 31) / 32), dim3(1024), 0, s, dw, v, g, rows, c, dg, coef);
  TG_CHECK_LAUNCH("wn_bwd_cols");
  hipLaunchKernelGGL(wn_bwd_apply, dim3(ew_grid((int64_t)rows * c)), dim3(256), 0, s, dw, v, coef, (int64_t)rows * c, c, dv);
  TG_CHECK_LAUNCH("wn_bwd_apply");
  return TG_OK;
}

}  // extern "C"
