// HBM-bound statistics / normalisation kernels.
//  * fused paths (what the models run): mean-only batch norm (apply after tg_igemm_colsum_*, backward statistics + apply) and
//    training-mode batch norm (statistics + apply, forward and backward), per application segment; statistics are per-workgroup
//    partial sums -> one fp64 atomic per column into one of REPL replicas of the accumulator (the fp64 sum of fp32 terms is
//    order-insensitive to ~1e-16, so replays are bit-identical after the cast to float);
//  * generic building blocks (shapes the fused paths do not take, evaluation mode): per-(segment, channel) column sums with a
//    deterministic two-stage reduction, finalisers, shift/scale + activation passes.
//
// Access pattern: rows are NHWC pixels, channels contiguous; every lane moves 16 B (float4), a block's 32
// column-groups cover 512 contiguous bytes per row, 8 rows in flight per pass.
#include "tg_common.h"
#include "tg_device.h"

namespace {

constexpr int RCH = 256;         // rows per stage-1 chunk
constexpr int MAXSEG = 8;
constexpr int REPL = 8;          // replicas of a statistics accumulator: workgroup b adds to replica b % REPL, the consumer sums them
                                 // (same-address fp64 atomics serialise at ~40 ns; 2 048 workgroups -> 64 deep per replica instead of 512)

struct SegTable { int nseg; int rows[MAXSEG]; };

__device__ __forceinline__ int seg_of_row(const SegTable& st, int r, int* row_in_seg_begin) {
  int b = 0;
  for (int s = 0; s < st.nseg; ++s) {
    if (r < b + st.rows[s]) { *row_in_seg_begin = b; return s; }
    b += st.rows[s];
  }
  *row_in_seg_begin = b;
  return st.nseg - 1;
}

// Row chunks enumerated segment by segment: workgroup `bid` gets rows [r0, r1) of ONE segment, whatever the segment sizes.
__device__ __forceinline__ bool bn_chunk(const SegTable& st, int chunk, int bid, int* seg, int* r0, int* r1) {
  int base = 0;
  for (int s = 0; s < st.nseg; ++s) {
    const int n = (st.rows[s] + chunk - 1) / chunk;
    if (bid < n) {
      *seg = s;
      *r0 = base + bid * chunk;
      *r1 = min(base + st.rows[s], *r0 + chunk);
      return true;
    }
    bid -= n;
    base += st.rows[s];
  }
  return false;
}

// mode: 0 SUM(a) ; 1 SUM(a), SUM(a^2) ; 2 SUM(a*act'(b)) ; 3 SUM(a), SUM(a*b) ;
// 4 SUM((a-mu)^2) with mu[c] = b[c]*alpha (b = per-channel sums of a previous mode-0 pass, alpha = 1/rows)
struct d4 { double x, y, z, w; };

// Partial sums are accumulated in fp64: batch-norm / mean-only-BN backward statistics are sums of mixed-sign terms
// that cancel to a small remainder, and the bias gradients behind them inherit that remainder coherently over all
// rows — fp32 accumulation costs ~1e-2 relative error there.  The kernel stays HBM-bound (8 fp64 adds per 16 B).
template <int MODE>
__global__ void __launch_bounds__(256) colstats_stage1(const float* __restrict__ a, const float* __restrict__ b, int ld_a, int ld_b,
                                                        int c4, SegTable st, int act, float alpha, double* __restrict__ part, int c_pad) {
  // locate this block's chunk: (segment, row range)
  int ch = blockIdx.y, seg = 0, base = 0;
  for (; seg < st.nseg; ++seg) {
    int n = (st.rows[seg] + RCH - 1) / RCH;
    if (ch < n) break;
    ch -= n;
    base += st.rows[seg];
  }
  if (seg >= st.nseg) return;
  const int r0 = base + ch * RCH;
  const int r1 = min(base + st.rows[seg], r0 + RCH);
  const int cg = blockIdx.x * 32 + (threadIdx.x & 31);   // float4 column group
  const int ry = threadIdx.x >> 5;
  d4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  if (cg < c4) {
    float4 mu = {0, 0, 0, 0};
    if (MODE == 4) {
      mu = *reinterpret_cast<const float4*>(b + cg * 4);
      mu.x *= alpha; mu.y *= alpha; mu.z *= alpha; mu.w *= alpha;
    }
    for (int r = r0 + ry; r < r1; r += 8) {
      const float4 va = *reinterpret_cast<const float4*>(a + (int64_t)r * ld_a + cg * 4);
      if (MODE == 0) {
        s1.x += va.x; s1.y += va.y; s1.z += va.z; s1.w += va.w;
      } else if (MODE == 1) {
        s1.x += va.x; s1.y += va.y; s1.z += va.z; s1.w += va.w;
        s2.x += (double)va.x * va.x; s2.y += (double)va.y * va.y; s2.z += (double)va.z * va.z; s2.w += (double)va.w * va.w;
      } else if (MODE == 4) {
        const float d0 = va.x - mu.x, d1 = va.y - mu.y, d2 = va.z - mu.z, d3 = va.w - mu.w;
        s1.x += (double)d0 * d0; s1.y += (double)d1 * d1; s1.z += (double)d2 * d2; s1.w += (double)d3 * d3;
      } else {
        const float4 vb = *reinterpret_cast<const float4*>(b + (int64_t)r * ld_b + cg * 4);
        if (MODE == 2) {
          s1.x += va.x * tgd::act_grad(vb.x, act, alpha); s1.y += va.y * tgd::act_grad(vb.y, act, alpha);
          s1.z += va.z * tgd::act_grad(vb.z, act, alpha); s1.w += va.w * tgd::act_grad(vb.w, act, alpha);
        } else {
          s1.x += va.x; s1.y += va.y; s1.z += va.z; s1.w += va.w;
          s2.x += (double)va.x * vb.x; s2.y += (double)va.y * vb.y; s2.z += (double)va.z * vb.z; s2.w += (double)va.w * vb.w;
        }
      }
    }
  }
  __shared__ d4 red[2][8][32];
  red[0][ry][threadIdx.x & 31] = s1;
  red[1][ry][threadIdx.x & 31] = s2;
  __syncthreads();
  if (ry == 0 && cg < c4) {
    for (int k = 1; k < 8; ++k) {
      const d4 t1 = red[0][k][threadIdx.x & 31], t2 = red[1][k][threadIdx.x & 31];
      s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
      s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
    }
    double* o = part + ((int64_t)blockIdx.y * 2) * c_pad + cg * 4;
    o[0] = s1.x; o[1] = s1.y; o[2] = s1.z; o[3] = s1.w;
    o[c_pad + 0] = s2.x; o[c_pad + 1] = s2.y; o[c_pad + 2] = s2.z; o[c_pad + 3] = s2.w;
  }
}

// stage 2: 32 columns x 8 chunk-lanes per block; lane l sums chunks l, l+8, ... (fixed order => reproducible), then an
// LDS tree.  (A single thread per column walking all ~1000 chunks is a 30 us dependent-load chain.)
__global__ void __launch_bounds__(256) colstats_stage2(const double* __restrict__ part, SegTable st, int c_pad, int c, float* __restrict__ s1,
                                                        float* __restrict__ s2) {
  const int seg = blockIdx.y;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  int ch0 = 0;
  for (int s = 0; s < seg; ++s) ch0 += (st.rows[s] + RCH - 1) / RCH;
  const int n = (st.rows[seg] + RCH - 1) / RCH;
  double a1 = 0., a2 = 0.;
  if (col < c) {
    for (int k = ty; k < n; k += 8) {
      a1 += part[((int64_t)(ch0 + k) * 2) * c_pad + col];
      a2 += part[((int64_t)(ch0 + k) * 2 + 1) * c_pad + col];
    }
  }
  __shared__ double red[2][8][32];
  red[0][ty][tx] = a1;
  red[1][ty][tx] = a2;
  __syncthreads();
  if (ty == 0 && col < c) {
    for (int k = 1; k < 8; ++k) { a1 += red[0][k][tx]; a2 += red[1][k][tx]; }
    s1[seg * c + col] = (float)a1;
    if (s2) s2[seg * c + col] = (float)a2;
  }
}

// y[r][c] = act(x[r][c]*scale[c] + shift[seg(r)][c]) for c < C; zero for C <= c < c_zero_to.
__global__ void __launch_bounds__(256) seg_scale_shift_act(const float* __restrict__ x, int ld_x, float* __restrict__ y, int ld_y, int rows, int c,
                                                           int c_zero_to, SegTable st, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, float alpha) {
  const int c4 = (c_zero_to + 3) / 4;
  const int64_t total = (int64_t)rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), cg = (int)(i - (int64_t)r * c4);
    int sb;
    const int seg = seg_of_row(st, r, &sb);
    float v[4];
    float4 xv = {0, 0, 0, 0};
    if (cg * 4 < c) xv = *reinterpret_cast<const float4*>(x + (int64_t)r * ld_x + cg * 4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = cg * 4 + k;
      float t = 0.f;
      if (cc < c) {
        t = xs[k];
        if (scale) t *= scale[cc];
        t = tgd::act(t + shift[seg * c + cc], act, alpha);
      }
      v[k] = t;
    }
    *reinterpret_cast<float4*>(y + (int64_t)r * ld_y + cg * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// dx[r][c] = dy[r][c]*act'(yact[r][c]) + shift[seg(r)][c]   (mean-only BN backward)
__global__ void __launch_bounds__(256) seg_actgrad_shift(const float* __restrict__ dy, int ld_dy, const float* __restrict__ yact, int ld_y,
                                                         float* __restrict__ dx, int ld_dx, int rows, int c, SegTable st,
                                                         const float* __restrict__ shift, int act, float alpha) {
  const int c4 = (c + 3) / 4;
  const int64_t total = (int64_t)rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), cg = (int)(i - (int64_t)r * c4);
    int sb;
    const int seg = seg_of_row(st, r, &sb);
    float4 g = *reinterpret_cast<const float4*>(dy + (int64_t)r * ld_dy + cg * 4);
    float4 ya = *reinterpret_cast<const float4*>(yact + (int64_t)r * ld_y + cg * 4);
    const float gs[4] = {g.x, g.y, g.z, g.w}, ys[4] = {ya.x, ya.y, ya.z, ya.w};
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = cg * 4 + k;
      v[k] = cc < c ? gs[k] * tgd::act_grad(ys[k], act, alpha) + shift[seg * c + cc] : 0.f;
    }
    *reinterpret_cast<float4*>(dx + (int64_t)r * ld_dx + cg * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// dx = (A[c]*dy + B[c]*x + C[c]) * (x > 0)     (batch-norm backward through the preceding ReLU; x = relu output)
__global__ void __launch_bounds__(256) bn_bwd_apply(const float* __restrict__ dy, int ld_dy, const float* __restrict__ x, int ld_x,
                                                    float* __restrict__ dx, int ld_dx, int rows, int c, const float* __restrict__ abc,
                                                    int relu_mask) {
  const int c4 = (c + 3) / 4;
  const int64_t total = (int64_t)rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), cg = (int)(i - (int64_t)r * c4);
    float4 g = *reinterpret_cast<const float4*>(dy + (int64_t)r * ld_dy + cg * 4);
    float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)r * ld_x + cg * 4);
    const float gs[4] = {g.x, g.y, g.z, g.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = cg * 4 + k;
      float t = 0.f;
      if (cc < c) {
        t = abc[cc] * gs[k] + abc[c + cc] * xs[k] + abc[2 * c + cc];
        if (relu_mask && !(xs[k] > 0.f)) t = 0.f;
      }
      v[k] = t;
    }
    *reinterpret_cast<float4*>(dx + (int64_t)r * ld_dx + cg * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// Fused mean-only-BN apply (Model/nn.py:147-187) on the output of tg_igemm_colsum_f32: each workgroup owns 32 consecutive rows
// (inside one application segment), derives shift[c] = b[c] - sums[seg][c]/rows_seg (training) or b[c] - pop_mean[c]
// (evaluation) once into LDS, then streams y = act(x + shift) in place with 16-B lanes.  Workgroup 0 also applies the
// sequential pop_mean updates.  Replaces stage-2 reduction + finalize + apply (three launches, one extra read pass).
__global__ void __launch_bounds__(256) mobn_apply(float* __restrict__ x, int ld, int rows, int c, SegTable st, const double* __restrict__ sums,
                                                  const float* __restrict__ b, float* __restrict__ pop, float decay, int act, float alpha) {
  __shared__ float shift[512];
  int seg, r0, r1;
  if (!bn_chunk(st, 32, blockIdx.x, &seg, &r0, &r1)) return;
  for (int k = threadIdx.x; k < c; k += 256) {
    const float bb = b ? b[k] : 0.f;
    shift[k] = sums ? bb - (float)(sums[(int64_t)seg * c + k] / (double)st.rows[seg]) : bb - pop[k];
  }
  if (blockIdx.x == 0 && sums) {
    for (int k = threadIdx.x; k < c; k += 256) {
      float pm = pop[k];
      for (int s = 0; s < st.nseg; ++s) pm = pm * decay + (float)(sums[(int64_t)s * c + k] / (double)st.rows[s]) * (1.f - decay);
      pop[k] = pm;
    }
  }
  __syncthreads();
  const int c4 = c >> 2;
  const int rows_here = r1 - r0;
  for (int i = threadIdx.x; i < rows_here * c4; i += 256) {
    const int rr = i / c4, cg = i - rr * c4;
    float4* p = reinterpret_cast<float4*>(x + (int64_t)(r0 + rr) * ld + cg * 4);
    float4 v = *p;
    v.x = tgd::act(v.x + shift[cg * 4], act, alpha);
    v.y = tgd::act(v.y + shift[cg * 4 + 1], act, alpha);
    v.z = tgd::act(v.z + shift[cg * 4 + 2], act, alpha);
    v.w = tgd::act(v.w + shift[cg * 4 + 3], act, alpha);
    *p = v;
  }
}

// mobn_apply + tf.nn.max_pool 2x2 + dropout behind it (Model/Good_GAN_cifar10.py:121-124,140-143) in ONE pass over the convolution's raw output:
// x = act(x + shift) in place (kept: the backward pass reads it) and pooled[n, h/2, w/2, c] = max over the 2x2 window of it, times the keep-mask and
// 1/keep.  A workgroup owns one pooled row of one image (two image rows: inside one application segment); workgroup 0 applies the sequential
// pop_mean updates as in mobn_apply.  Saves the max-pool launch's read of the activated tensor.
__global__ void __launch_bounds__(256) mobn_apply_pool(float* __restrict__ x, int ld, int h, int w, int c, SegTable st, const double* __restrict__ sums,
                                                       const float* __restrict__ b, float* __restrict__ pop, float decay, int act, float alpha,
                                                       float* __restrict__ out, int ld_out, const float* __restrict__ mask, int ld_m, float mscale) {
  __shared__ float shift[512];
  const int ho = h >> 1, wo = w >> 1;
  const int img = blockIdx.x / ho, oy = blockIdx.x - img * ho;
  const int64_t row0 = ((int64_t)img * h + 2 * oy) * w;                 // first pixel row of this workgroup
  int seg = 0;
  int64_t acc_rows = st.rows[0];
  while (seg < st.nseg - 1 && row0 >= acc_rows) acc_rows += st.rows[++seg];
  for (int k = threadIdx.x; k < c; k += 256) {
    const float bb = b ? b[k] : 0.f;
    shift[k] = sums ? bb - (float)(sums[(int64_t)seg * c + k] / (double)st.rows[seg]) : bb - pop[k];
  }
  if (blockIdx.x == 0 && sums) {
    for (int k = threadIdx.x; k < c; k += 256) {
      float pm = pop[k];
      for (int s = 0; s < st.nseg; ++s) pm = pm * decay + (float)(sums[(int64_t)s * c + k] / (double)st.rows[s]) * (1.f - decay);
      pop[k] = pm;
    }
  }
  __syncthreads();
  const int c4 = c >> 2;
  for (int i = threadIdx.x; i < wo * c4; i += 256) {
    const int ox = i / c4, cg = i - ox * c4;
    float* p = x + (row0 + 2 * ox) * ld + cg * 4;
    float4 v[4] = {*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + ld), *reinterpret_cast<const float4*>(p + (int64_t)w * ld),
                   *reinterpret_cast<const float4*>(p + (int64_t)(w + 1) * ld)};
    const float s0 = shift[cg * 4], s1 = shift[cg * 4 + 1], s2 = shift[cg * 4 + 2], s3 = shift[cg * 4 + 3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q].x = tgd::act(v[q].x + s0, act, alpha); v[q].y = tgd::act(v[q].y + s1, act, alpha);
      v[q].z = tgd::act(v[q].z + s2, act, alpha); v[q].w = tgd::act(v[q].w + s3, act, alpha);
    }
    *reinterpret_cast<float4*>(p) = v[0];
    *reinterpret_cast<float4*>(p + ld) = v[1];
    *reinterpret_cast<float4*>(p + (int64_t)w * ld) = v[2];
    *reinterpret_cast<float4*>(p + (int64_t)(w + 1) * ld) = v[3];
    float4 m = make_float4(fmaxf(fmaxf(v[0].x, v[1].x), fmaxf(v[2].x, v[3].x)), fmaxf(fmaxf(v[0].y, v[1].y), fmaxf(v[2].y, v[3].y)),
                           fmaxf(fmaxf(v[0].z, v[1].z), fmaxf(v[2].z, v[3].z)), fmaxf(fmaxf(v[0].w, v[1].w), fmaxf(v[2].w, v[3].w)));
    const int64_t pix = ((int64_t)img * ho + oy) * wo + ox;
    if (mask) {
      const float4 k = *reinterpret_cast<const float4*>(mask + pix * ld_m + cg * 4);
      m.x *= k.x * mscale; m.y *= k.y * mscale; m.z *= k.z * mscale; m.w *= k.w * mscale;
    }
    *reinterpret_cast<float4*>(out + pix * ld_out + cg * 4) = m;
  }
}

// Mean-only-BN backward in two launches: (1) sums[seg][c] += sum_rows dy*act'(y) (one fp64 atomic per column per workgroup),
// (2) dpre = dy*act'(y) - sums[seg]/rows_seg with the shift derived in LDS; workgroup 0 writes db = sum_seg sums.
__global__ void __launch_bounds__(256) mobn_bwd_sums(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y, int rows, int c,
                                                     SegTable st, int act, float alpha, double* __restrict__ sums, int chunk) {
  // each workgroup walks `chunk` rows of one segment (a few hundred workgroups in all: the atomics on the nseg*c accumulators
  // serialise, 9 600 32-row workgroups were slower than the two-stage reduction this replaces)
  int seg, r0, r1;
  if (!bn_chunk(st, chunk, blockIdx.x, &seg, &r0, &r1)) return;
  const int c4 = c >> 2;
  const int lanes = 256 / c4 > 0 ? 256 / c4 : 1;
  const int cg = threadIdx.x % c4, rl = threadIdx.x / c4;
  __shared__ float4 red[256];
  float4 acc = {0, 0, 0, 0};
  if (rl < lanes) {
    for (int r = r0 + rl; r < r1; r += lanes) {
      const float4 g = *reinterpret_cast<const float4*>(dy + (int64_t)r * ld_dy + cg * 4);
      const float4 yy = *reinterpret_cast<const float4*>(y + (int64_t)r * ld_y + cg * 4);
      acc.x += g.x * tgd::act_grad(yy.x, act, alpha); acc.y += g.y * tgd::act_grad(yy.y, act, alpha);
      acc.z += g.z * tgd::act_grad(yy.z, act, alpha); acc.w += g.w * tgd::act_grad(yy.w, act, alpha);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < lanes; ++k) {
      const float4 t = red[k * c4 + cg];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    double* o = sums + ((int64_t)(blockIdx.x % REPL) * st.nseg + seg) * c + cg * 4;
    atomicAdd(o, (double)acc.x); atomicAdd(o + 1, (double)acc.y); atomicAdd(o + 2, (double)acc.z); atomicAdd(o + 3, (double)acc.w);
  }
}

__device__ __forceinline__ double repl_sum(const double* __restrict__ sums, int nseg, int seg, int c, int k) {
  double t = 0.;
#pragma unroll
  for (int r = 0; r < REPL; ++r) t += sums[((int64_t)r * nseg + seg) * c + k];      // fixed order
  return t;
}

__global__ void __launch_bounds__(256) mobn_bwd_apply(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                      float* __restrict__ dx, int ld_dx, int rows, int c, SegTable st, int act, float alpha,
                                                      const double* __restrict__ sums, float* __restrict__ db) {
  __shared__ float shift[512];
  int seg, r0, r1;
  if (!bn_chunk(st, 32, blockIdx.x, &seg, &r0, &r1)) return;
  for (int k = threadIdx.x; k < c; k += 256) shift[k] = -(float)(repl_sum(sums, st.nseg, seg, c, k) / (double)st.rows[seg]);
  if (blockIdx.x == 0 && db) {
    for (int k = threadIdx.x; k < c; k += 256) {
      double t = 0.;
      for (int s = 0; s < st.nseg; ++s) t += repl_sum(sums, st.nseg, s, c, k);
      db[k] = (float)t;
    }
  }
  __syncthreads();
  const int c4 = c >> 2;
  const int rows_here = r1 - r0;
  for (int i = threadIdx.x; i < rows_here * c4; i += 256) {
    const int rr = i / c4, cg = i - rr * c4;
    const float4 g = *reinterpret_cast<const float4*>(dy + (int64_t)(r0 + rr) * ld_dy + cg * 4);
    const float4 yy = *reinterpret_cast<const float4*>(y + (int64_t)(r0 + rr) * ld_y + cg * 4);
    float4 o;
    o.x = g.x * tgd::act_grad(yy.x, act, alpha) + shift[cg * 4];
    o.y = g.y * tgd::act_grad(yy.y, act, alpha) + shift[cg * 4 + 1];
    o.z = g.z * tgd::act_grad(yy.z, act, alpha) + shift[cg * 4 + 2];
    o.w = g.w * tgd::act_grad(yy.w, act, alpha) + shift[cg * 4 + 3];
    *reinterpret_cast<float4*>(dx + (int64_t)(r0 + rr) * ld_dx + cg * 4) = o;
  }
}

// dx = t - mean_seg(t) given the per-segment column sums of t (single accumulator copy, written by tg_igemm_actsum_*); db = sum of sums
__global__ void __launch_bounds__(256) mobn_center(const float* __restrict__ t, int ld_t, float* __restrict__ dx, int ld_dx, int c, SegTable st,
                                                   const double* __restrict__ sums, int n_repl, float* __restrict__ db) {
  __shared__ float shift[512];
  int seg, r0, r1;
  if (!bn_chunk(st, 32, blockIdx.x, &seg, &r0, &r1)) return;
  auto total = [&](int s, int k) {
    double a = 0.;
    for (int r = 0; r < n_repl; ++r) a += sums[((int64_t)r * st.nseg + s) * c + k];      // fixed order
    return a;
  };
  for (int k = threadIdx.x; k < c; k += 256) shift[k] = -(float)(total(seg, k) / (double)st.rows[seg]);
  if (blockIdx.x == 0 && db) {
    for (int k = threadIdx.x; k < c; k += 256) {
      double a = 0.;
      for (int s = 0; s < st.nseg; ++s) a += total(s, k);
      db[k] = (float)a;
    }
  }
  __syncthreads();
  const int c4 = c >> 2;
  const int rows_here = r1 - r0;
  for (int i = threadIdx.x; i < rows_here * c4; i += 256) {
    const int rr = i / c4, cg = i - rr * c4;
    float4 v = *reinterpret_cast<const float4*>(t + (int64_t)(r0 + rr) * ld_t + cg * 4);
    v.x += shift[cg * 4]; v.y += shift[cg * 4 + 1]; v.z += shift[cg * 4 + 2]; v.w += shift[cg * 4 + 3];
    *reinterpret_cast<float4*>(dx + (int64_t)(r0 + rr) * ld_dx + cg * 4) = v;
  }
}

// 2x2 max-pool (+ dropout) backward whose result feeds a mean-only-BN layer: routes the pooled gradient to the arg-max position,
// multiplies by act'(y) and accumulates the per-application column sums of the product t — the statistics pass of that layer's
// backward — in the same pass.  Work unit: pooled pixels; `st` holds the application sizes in POOLED pixels.
__global__ void __launch_bounds__(256) maxpool2_bwd_actsum(const float* __restrict__ dout, int ld_do, const float* __restrict__ mask, int ld_m,
                                                           float mscale, const float* __restrict__ y, int ld_y, float* __restrict__ t, int ld_t,
                                                           int h, int w, int c, SegTable st, int chunk, int act, float alpha,
                                                           double* __restrict__ sums) {
  int seg, r0, r1;
  if (!bn_chunk(st, chunk, blockIdx.x, &seg, &r0, &r1)) return;
  const int ho = h >> 1, wo = w >> 1;
  const int c4 = c >> 2;
  const int lanes = 256 / c4 > 0 ? 256 / c4 : 1;
  const int cg = threadIdx.x % c4, rl = threadIdx.x / c4;
  __shared__ float4 red[256];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < lanes) {
    for (int pp = r0 + rl; pp < r1; pp += lanes) {
      const int ox = pp % wo, oy = (pp / wo) % ho, img = pp / (wo * ho);
      const int64_t base = ((int64_t)img * h + 2 * oy) * w + 2 * ox;
      const int64_t pos[4] = {base, base + 1, base + w, base + w + 1};
      float yv[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(y + pos[q] * ld_y + cg * 4);
        yv[q][0] = v.x; yv[q][1] = v.y; yv[q][2] = v.z; yv[q][3] = v.w;
      }
      const float4 gv = *reinterpret_cast<const float4*>(dout + (int64_t)pp * ld_do + cg * 4);
      float g[4] = {gv.x, gv.y, gv.z, gv.w};
      if (mask) {
        const float4 m = *reinterpret_cast<const float4*>(mask + (int64_t)pp * ld_m + cg * 4);
        g[0] *= m.x * mscale; g[1] *= m.y * mscale; g[2] *= m.z * mscale; g[3] *= m.w * mscale;
      }
      float o[4][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int am = 0;
        float mv = yv[0][e];
        if (yv[1][e] > mv) { mv = yv[1][e]; am = 1; }
        if (yv[2][e] > mv) { mv = yv[2][e]; am = 2; }
        if (yv[3][e] > mv) { mv = yv[3][e]; am = 3; }
        const float tv = g[e] * tgd::act_grad(mv, act, alpha);
        acc[e] += tv;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][e] = q == am ? tv : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(t + pos[q] * ld_t + cg * 4) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
    }
  }
  red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < lanes; ++k) {
      const float4 v = red[k * c4 + cg];
      acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
    }
    double* o = sums + ((int64_t)(blockIdx.x % REPL) * st.nseg + seg) * c + cg * 4;
    atomicAdd(o, (double)acc[0]); atomicAdd(o + 1, (double)acc[1]); atomicAdd(o + 2, (double)acc[2]); atomicAdd(o + 3, (double)acc[3]);
  }
}

// dpre = dy * act'(yact) (pad columns zeroed up to ld_out) AND its column sums (the bias gradient of a plain conv / transposed conv /
// dense layer) in one pass: per-workgroup partial sums -> one fp64 atomic per column into replica (workgroup % REPL); a second tiny
// launch adds the replicas.  Replaces actgrad + two-stage colstats (three launches, two passes over dpre).
__global__ void __launch_bounds__(256) actgrad_bias(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                    float* __restrict__ out, int ld_out, int rows, int c, int act, float alpha, int chunk,
                                                    double* __restrict__ sums) {
  const int r0 = blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
  const int g4 = ld_out >> 2;                         // column groups of the output row (covers the padding)
  const int lanes = 256 / g4 > 0 ? 256 / g4 : 1;
  const int cg = threadIdx.x % g4, rl = threadIdx.x / g4;
  __shared__ float4 red[256];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (c & 3) == 0 && (ld_dy & 3) == 0 && (!y || (ld_y & 3) == 0);
  if (rl < lanes) {
    for (int r = r0 + rl; r < r1; r += lanes) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        if (cg * 4 < c) {
          const float4 g = *reinterpret_cast<const float4*>(dy + (int64_t)r * ld_dy + cg * 4);
          v[0] = g.x; v[1] = g.y; v[2] = g.z; v[3] = g.w;
          if (y) {
            const float4 yy = *reinterpret_cast<const float4*>(y + (int64_t)r * ld_y + cg * 4);
            v[0] *= tgd::act_grad(yy.x, act, alpha); v[1] *= tgd::act_grad(yy.y, act, alpha);
            v[2] *= tgd::act_grad(yy.z, act, alpha); v[3] *= tgd::act_grad(yy.w, act, alpha);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int col = cg * 4 + k;
          if (col < c) {
            float t = dy[(int64_t)r * ld_dy + col];
            if (y) t *= tgd::act_grad(y[(int64_t)r * ld_y + col], act, alpha);
            v[k] = t;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += v[k];
      *reinterpret_cast<float4*>(out + (int64_t)r * ld_out + cg * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < lanes; ++k) {
      const float4 t = red[k * g4 + cg];
      acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (cg * 4 + k < c) atomicAdd(sums + (int64_t)(blockIdx.x % REPL) * c + cg * 4 + k, (double)acc[k]);
  }
}

__global__ void repl_finalize(const double* __restrict__ sums, int c, float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= c) return;
  double t = 0.;
#pragma unroll
  for (int r = 0; r < REPL; ++r) t += sums[(int64_t)r * c + k];
  out[k] = (float)t;
}

__global__ void mobn_finalize(const float* __restrict__ sums, SegTable st, int c, const float* __restrict__ b, float* __restrict__ pop, float decay,
                              int train, float* __restrict__ shift) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  float pm = pop[col];
  const float bb = b ? b[col] : 0.f;
  for (int s = 0; s < st.nseg; ++s) {
    if (train) {
      const float m = sums[s * c + col] / (float)st.rows[s];
      shift[s * c + col] = bb - m;
      pm = pm * decay + m * (1.f - decay);   // sequential pop_mean updates in call-site order (nn.py:181)
    } else {
      shift[s * c + col] = bb - pm;
    }
  }
  if (train) pop[col] = pm;
}

__global__ void mobn_bwd_finalize(const float* __restrict__ sums, SegTable st, int c, float* __restrict__ shift, float* __restrict__ db) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  float tot = 0.f;
  for (int s = 0; s < st.nseg; ++s) {
    const float v = sums[s * c + col];
    shift[s * c + col] = -v / (float)st.rows[s];
    tot += v;
  }
  db[col] = tot;
}

__global__ void bn_finalize(const float* __restrict__ s1, const float* __restrict__ s2, int rows, int c, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float eps, float* __restrict__ scale, float* __restrict__ shift,
                            float* __restrict__ mean_inv, float* __restrict__ mm, float* __restrict__ mv, float decay, int bessel) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  const float mu = s1[col] / (float)rows;
  const float var = s2[col] / (float)rows;       // biased; s2 = sum (x-mu)^2 from the second (centred) pass, as tf.nn.moments
  const float inv = 1.f / sqrtf(var + eps);
  const float sc = gamma[col] * inv;
  scale[col] = sc;
  shift[col] = beta[col] - mu * sc;
  mean_inv[col] = mu;
  mean_inv[c + col] = inv;
  if (mm) {   // moving statistics: dead state of G's always-training BN
    const float v = bessel && rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
    mm[col] = mm[col] * decay + mu * (1.f - decay);
    mv[col] = mv[col] * decay + v * (1.f - decay);
  }
}

__global__ void bn_eval_finalize(int c, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mm,
                                 const float* __restrict__ mv, float eps, float* __restrict__ scale, float* __restrict__ shift) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  const float sc = gamma[col] / sqrtf(mv[col] + eps);
  scale[col] = sc;
  shift[col] = beta[col] - mm[col] * sc;
}

__global__ void bn_bwd_finalize(const float* __restrict__ s_dy, const float* __restrict__ s_dyx, int rows, int c, const float* __restrict__ gamma,
                                const float* __restrict__ mean_inv, float* __restrict__ abc, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  const float mu = mean_inv[col], inv = mean_inv[c + col], g = gamma[col];
  const float dbt = s_dy[col];
  const float dgm = inv * (s_dyx[col] - mu * dbt);     // sum dy * xhat
  dgamma[col] = dgm;
  dbeta[col] = dbt;
  const float m = (float)rows;
  const float A = g * inv;
  const float B = -g * inv * inv * dgm / m;
  abc[col] = A;
  abc[c + col] = B;
  abc[2 * c + col] = -A * dbt / m - B * mu;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused training-mode batch norm (tf.contrib.layers.batch_norm, Model/modle_base.py:229-237) over application segments:
// two launches per direction instead of six / four.
//   sums   : per (segment, column) fp64 accumulators filled with one atomic per column per workgroup;
//            forward  S0 = sum x, S1 = sum x^2   (variance = S1/n - mean^2 evaluated in fp64: no cancellation problem)
//            backward S0 = sum dy, S1 = sum dy*x
//   apply  : every workgroup derives the per-column coefficients of its segment in LDS and streams its rows;
//            the first workgroup of a column block also writes mean / inv-std (forward) or dgamma / dbeta (backward)
//            and updates the moving statistics, sequentially over the segments (= call-site order of the applications).
// Work decomposition: blockIdx.x enumerates row chunks segment by segment (a chunk never straddles two segments, whatever
// their sizes), blockIdx.y column blocks of BN_CW columns.
constexpr int BN_CW = 256;

template <bool BWD>
__global__ void __launch_bounds__(256) bn_sums(const float* __restrict__ a, int ld_a, const float* __restrict__ b, int ld_b, int c, SegTable st,
                                               int chunk, double* __restrict__ sums) {
  int seg, r0, r1;
  if (!bn_chunk(st, chunk, blockIdx.x, &seg, &r0, &r1)) return;
  const int c0 = blockIdx.y * BN_CW;
  const int cgn = min((c - c0 + 3) / 4, BN_CW / 4);          // column groups of this block
  const int lanes = 256 / cgn;
  const int cg = threadIdx.x % cgn, rl = threadIdx.x / cgn;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (rl < lanes) {
    for (int r = r0 + rl; r < r1; r += lanes) {
      const float4 av = *reinterpret_cast<const float4*>(a + (int64_t)r * ld_a + c0 + cg * 4);
      const float as[4] = {av.x, av.y, av.z, av.w};
      if (BWD) {
        const float4 bv = *reinterpret_cast<const float4*>(b + (int64_t)r * ld_b + c0 + cg * 4);
        const float bs[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += (double)as[k]; s1[k] += (double)as[k] * (double)bs[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += (double)as[k]; s1[k] += (double)as[k] * (double)as[k]; }
      }
    }
  }
  __shared__ double red[256 * 8];
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[threadIdx.x * 8 + k] = s0[k]; red[threadIdx.x * 8 + 4 + k] = s1[k]; }
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < lanes; ++l)
#pragma unroll
      for (int k = 0; k < 4; ++k) { s0[k] += red[(l * cgn + cg) * 8 + k]; s1[k] += red[(l * cgn + cg) * 8 + 4 + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = c0 + cg * 4 + k;
      if (col < c) {
        double* o = sums + (((int64_t)(blockIdx.x % REPL) * st.nseg + seg) * 2) * c + col;
        atomicAdd(o, s0[k]);
        atomicAdd(o + c, s1[k]);
      }
    }
  }
}

// S0 / S1 of (segment, column) summed over the replicas (layout [REPL][nseg][2][c]); fixed order
__device__ __forceinline__ void bn_repl(const double* __restrict__ sums, int nseg, int seg, int c, int col, double* s0, double* s1) {
  double a = 0., b = 0.;
#pragma unroll
  for (int r = 0; r < REPL; ++r) {
    const double* o = sums + (((int64_t)r * nseg + seg) * 2) * c + col;
    a += o[0]; b += o[c];
  }
  *s0 = a; *s1 = b;
}

__global__ void __launch_bounds__(256) bn_train_apply(const float* __restrict__ x, int ld_x, float* __restrict__ y, int ld_y, int c, SegTable st, int chunk,
                                                      const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float decay, float* __restrict__ mm, float* __restrict__ mv,
                                                      float* __restrict__ mean_inv) {
  int seg, r0, r1;
  if (!bn_chunk(st, chunk, blockIdx.x, &seg, &r0, &r1)) return;
  const int c0 = blockIdx.y * BN_CW;
  const int ncol = min(c - c0, BN_CW);
  __shared__ float sc[BN_CW], sh[BN_CW];
  for (int k = threadIdx.x; k < BN_CW; k += 256) {
    float a = 0.f, b = 0.f;
    if (k < ncol) {
      const int col = c0 + k;
      const double n = (double)st.rows[seg];
      double q0, q1;
      bn_repl(sums, st.nseg, seg, c, col, &q0, &q1);
      const double mu = q0 / n;
      double var = q1 / n - mu * mu;
      var = var > 0. ? var : 0.;
      const float inv = 1.f / sqrtf((float)var + eps);
      a = gamma[col] * inv;
      b = beta[col] - (float)mu * a;
    }
    sc[k] = a; sh[k] = b;
  }
  // first workgroup of this column block: per-segment mean / inv-std for the backward pass, moving statistics in segment order
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < ncol; k += 256) {
      const int col = c0 + k;
      float m_run = mm ? mm[col] : 0.f, v_run = mv ? mv[col] : 0.f;
      for (int s = 0; s < st.nseg; ++s) {
        const double n = (double)st.rows[s];
        double q0, q1;
        bn_repl(sums, st.nseg, s, c, col, &q0, &q1);
        const double mu = q0 / n;
        double var = q1 / n - mu * mu;
        var = var > 0. ? var : 0.;
        mean_inv[((int64_t)s * 2) * c + col] = (float)mu;
        mean_inv[((int64_t)s * 2 + 1) * c + col] = 1.f / sqrtf((float)var + eps);
        const float vb = st.rows[s] > 1 ? (float)var * ((float)st.rows[s] / (float)(st.rows[s] - 1)) : (float)var;   // Bessel, as the fused TF op
        m_run = m_run * decay + (float)mu * (1.f - decay);
        v_run = v_run * decay + vb * (1.f - decay);
      }
      if (mm) { mm[col] = m_run; mv[col] = v_run; }
    }
  }
  __syncthreads();
  const int cgn = (ncol + 3) / 4;
  const int total = (r1 - r0) * cgn;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int rr = i / cgn, cg = i - rr * cgn;
    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + rr) * ld_x + c0 + cg * 4);
    float4 o;
    o.x = xv.x * sc[cg * 4] + sh[cg * 4];
    o.y = xv.y * sc[cg * 4 + 1] + sh[cg * 4 + 1];
    o.z = xv.z * sc[cg * 4 + 2] + sh[cg * 4 + 2];
    o.w = xv.w * sc[cg * 4 + 3] + sh[cg * 4 + 3];
    *reinterpret_cast<float4*>(y + (int64_t)(r0 + rr) * ld_y + c0 + cg * 4) = o;
  }
}

// the moving-statistics chain of bn_train_apply alone (same arithmetic, same order), from the sums a forward launch left behind
__global__ void __launch_bounds__(256) bn_moving_update(const double* __restrict__ sums, int c, SegTable st, float decay, float* __restrict__ mm,
                                                        float* __restrict__ mv) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  float m_run = mm[col], v_run = mv[col];
  for (int s = 0; s < st.nseg; ++s) {
    const double n = (double)st.rows[s];
    double q0, q1;
    bn_repl(sums, st.nseg, s, c, col, &q0, &q1);
    const double mu = q0 / n;
    double var = q1 / n - mu * mu;
    var = var > 0. ? var : 0.;
    const float vb = st.rows[s] > 1 ? (float)var * ((float)st.rows[s] / (float)(st.rows[s] - 1)) : (float)var;
    m_run = m_run * decay + (float)mu * (1.f - decay);
    v_run = v_run * decay + vb * (1.f - decay);
  }
  mm[col] = m_run; mv[col] = v_run;
}

__global__ void __launch_bounds__(256) bn_train_bwd_apply(const float* __restrict__ dy, int ld_dy, const float* __restrict__ x, int ld_x,
                                                          float* __restrict__ dx, int ld_dx, int c, SegTable st, int chunk,
                                                          const double* __restrict__ sums, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_inv, int act, float alpha, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, double* __restrict__ dsum) {
  // act != none: x is the output of a fused activation (relu / leaky relu); dx is then the gradient with respect to the PRE-activation
  // value: times act'(x).  dsum != NULL: the column sums of dx (= the bias gradient of the layer that produced x) are accumulated into
  // dsum[REPL][c] (fp64, one atomic per column per workgroup) — tg_actgrad_bias_f32's pass folded into this one.
  int seg, r0, r1;
  if (!bn_chunk(st, chunk, blockIdx.x, &seg, &r0, &r1)) return;
  const int c0 = blockIdx.y * BN_CW;
  const int ncol = min(c - c0, BN_CW);
  __shared__ float A[BN_CW], B[BN_CW], Cc[BN_CW];
  for (int k = threadIdx.x; k < BN_CW; k += 256) {
    float a = 0.f, b = 0.f, cc = 0.f;
    if (k < ncol) {
      const int col = c0 + k;
      const float mu = mean_inv[((int64_t)seg * 2) * c + col], inv = mean_inv[((int64_t)seg * 2 + 1) * c + col], g = gamma[col];
      const float m = (float)st.rows[seg];
      double q0, q1;
      bn_repl(sums, st.nseg, seg, c, col, &q0, &q1);
      const float dbt = (float)q0;
      const float dgm = inv * (float)(q1 - (double)mu * q0);   // sum dy * xhat
      a = g * inv;
      b = -g * inv * inv * dgm / m;
      cc = -a * dbt / m - b * mu;
    }
    A[k] = a; B[k] = b; Cc[k] = cc;
  }
  if (blockIdx.x == 0 && dgamma) {
    for (int k = threadIdx.x; k < ncol; k += 256) {
      const int col = c0 + k;
      double dg = 0., db = 0.;
      for (int s = 0; s < st.nseg; ++s) {
        const double mu = (double)mean_inv[((int64_t)s * 2) * c + col], inv = (double)mean_inv[((int64_t)s * 2 + 1) * c + col];
        double sdy, sdyx;
        bn_repl(sums, st.nseg, s, c, col, &sdy, &sdyx);
        db += sdy;
        dg += inv * (sdyx - mu * sdy);
      }
      dgamma[col] = (float)dg;
      dbeta[col] = (float)db;
    }
  }
  __syncthreads();
  const int cgn = (ncol + 3) / 4;
  const int total = (r1 - r0) * cgn;
  double cs[4] = {0., 0., 0., 0.};                             // dsum: this thread's column group is fixed (256 % cgn == 0, launcher check)
  for (int i = threadIdx.x; i < total; i += 256) {
    const int rr = i / cgn, cg = i - rr * cgn;
    const float4 g = *reinterpret_cast<const float4*>(dy + (int64_t)(r0 + rr) * ld_dy + c0 + cg * 4);
    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + rr) * ld_x + c0 + cg * 4);
    const float gs[4] = {g.x, g.y, g.z, g.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = A[cg * 4 + k] * gs[k] + B[cg * 4 + k] * xs[k] + Cc[cg * 4 + k];
      if (act != TG_ACT_NONE) t *= tgd::act_grad(xs[k], act, alpha);
      v[k] = t;
      cs[k] += (double)t;
    }
    *reinterpret_cast<float4*>(dx + (int64_t)(r0 + rr) * ld_dx + c0 + cg * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (dsum != nullptr) {
    __shared__ double red[256 * 4];
    const int cg = threadIdx.x % cgn, rl = threadIdx.x / cgn, lanes = 256 / cgn;
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = cs[k];
    __syncthreads();
    if (rl == 0) {
      for (int l = 1; l < lanes; ++l)
#pragma unroll
        for (int k = 0; k < 4; ++k) cs[k] += red[(l * cgn + cg) * 4 + k];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = c0 + cg * 4 + k;
        if (col < c) atomicAdd(dsum + (int64_t)(blockIdx.x % REPL) * c + col, cs[k]);
      }
    }
  }
}

int make_segs(SegTable& st, const int32_t* seg_rows, int nseg, int rows) {
  TG_REQUIRE(nseg >= 1 && nseg <= MAXSEG, "nseg=%d out of range", nseg);
  int tot = 0;
  st.nseg = nseg;
  for (int i = 0; i < nseg; ++i) { st.rows[i] = seg_rows[i]; tot += seg_rows[i]; TG_REQUIRE(seg_rows[i] > 0, "empty segment %d", i); }
  TG_REQUIRE(tot == rows, "segments sum to %d, rows=%d", tot, rows);
  return TG_OK;
}

int num_chunks(const SegTable& st) {
  int n = 0;
  for (int i = 0; i < st.nseg; ++i) n += (st.rows[i] + RCH - 1) / RCH;
  return n;
}

int ew_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" {

int64_t tg_colstats_workspace_floats(int rows, int nseg, int c) {
  // upper bound: every segment adds at most one partial chunk
  int c_pad = (c + 3) / 4 * 4;
  return (int64_t)((rows + RCH - 1) / RCH + nseg) * 2 * c_pad * 2;   // fp64 partials
}

int tg_colstats_f32(int mode, const float* a, int ld_a, const float* b, int ld_b, int rows, int c, const int32_t* seg_rows, int nseg,
                    int act, float alpha, float* workspace, float* s1, float* s2, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(a && workspace && s1, "colstats: null buffer");
  TG_REQUIRE(ld_a % 4 == 0 && (b == nullptr || ld_b % 4 == 0), "colstats: ld must be a multiple of 4");
  TG_REQUIRE(mode != 4 || nseg == 1, "colstats: mode 4 needs one segment");   // b must hold (c+3)/4*4 readable floats
  TG_REQUIRE((mode == 0 || mode == 1) || b != nullptr, "colstats: mode %d needs operand b", mode);
  const int c4 = (c + 3) / 4, c_pad = c4 * 4;
  TG_REQUIRE(c_pad <= ld_a && (b == nullptr || mode == 4 || c_pad <= ld_b), "colstats: c=%d exceeds ld", c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 4.0 * rows * c * (b ? 2 : 1), s);
  dim3 grid((c4 + 31) / 32, num_chunks(st));
  TG_REQUIRE((uintptr_t)workspace % 8 == 0, "colstats: workspace must be 8-byte aligned");
  double* wsd = reinterpret_cast<double*>(workspace);
  switch (mode) {
    case 0: hipLaunchKernelGGL(colstats_stage1<0>, grid, dim3(256), 0, s, a, b, ld_a, ld_b, c4, st, act, alpha, wsd, c_pad); break;
    case 1: hipLaunchKernelGGL(colstats_stage1<1>, grid, dim3(256), 0, s, a, b, ld_a, ld_b, c4, st, act, alpha, wsd, c_pad); break;
    case 2: hipLaunchKernelGGL(colstats_stage1<2>, grid, dim3(256), 0, s, a, b, ld_a, ld_b, c4, st, act, alpha, wsd, c_pad); break;
    case 3: hipLaunchKernelGGL(colstats_stage1<3>, grid, dim3(256), 0, s, a, b, ld_a, ld_b, c4, st, act, alpha, wsd, c_pad); break;
    case 4: hipLaunchKernelGGL(colstats_stage1<4>, grid, dim3(256), 0, s, a, b, ld_a, ld_b, c4, st, act, alpha, wsd, c_pad); break;
    default: tg::set_error("colstats: bad mode %d", mode); return TG_ERR_INVALID;
  }
  TG_CHECK_LAUNCH("colstats_stage1");
  hipLaunchKernelGGL(colstats_stage2, dim3((c + 31) / 32, nseg), dim3(256), 0, s, wsd, st, c_pad, c, s1, s2);
  TG_CHECK_LAUNCH("colstats_stage2");
  return TG_OK;
}

int tg_seg_scale_shift_act_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, int c_zero_to, const int32_t* seg_rows, int nseg,
                               const float* scale, const float* shift, int act, float alpha, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(x && y && shift, "seg_scale_shift_act: null buffer");
  TG_REQUIRE(ld_x % 4 == 0 && ld_y % 4 == 0 && c_zero_to >= c && (c_zero_to + 3) / 4 * 4 <= ld_y && (c + 3) / 4 * 4 <= ld_x,
             "seg_scale_shift_act: c=%d c_zero_to=%d ld_x=%d ld_y=%d", c, c_zero_to, ld_x, ld_y);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 8.0 * rows * c, s);
  hipLaunchKernelGGL(seg_scale_shift_act, dim3(ew_grid((int64_t)rows * ((c_zero_to + 3) / 4))), dim3(256), 0, s, x, ld_x, y, ld_y, rows, c,
                     c_zero_to, st, scale, shift, act, alpha);
  TG_CHECK_LAUNCH("seg_scale_shift_act");
  return TG_OK;
}

int tg_seg_actgrad_shift_f32(const float* dy, int ld_dy, const float* yact, int ld_y, float* dx, int ld_dx, int rows, int c,
                             const int32_t* seg_rows, int nseg, const float* shift, int act, float alpha, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(dy && yact && dx && shift, "seg_actgrad_shift: null buffer");
  const int cp = (c + 3) / 4 * 4;
  TG_REQUIRE(ld_dy % 4 == 0 && ld_y % 4 == 0 && ld_dx % 4 == 0 && cp <= ld_dy && cp <= ld_y && cp <= ld_dx, "seg_actgrad_shift: c=%d vs ld", c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 12.0 * rows * c, s);
  hipLaunchKernelGGL(seg_actgrad_shift, dim3(ew_grid((int64_t)rows * (cp / 4))), dim3(256), 0, s, dy, ld_dy, yact, ld_y, dx, ld_dx, rows, c, st,
                     shift, act, alpha);
  TG_CHECK_LAUNCH("seg_actgrad_shift");
  return TG_OK;
}

int tg_bn_bwd_apply_f32(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const float* abc,
                        int relu_mask, void* stream) {
  TG_REQUIRE(dy && x && dx && abc, "bn_bwd_apply: null buffer");
  const int cp = (c + 3) / 4 * 4;
  TG_REQUIRE(ld_dy % 4 == 0 && ld_x % 4 == 0 && ld_dx % 4 == 0 && cp <= ld_dy && cp <= ld_x && cp <= ld_dx, "bn_bwd_apply: c=%d vs ld", c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 12.0 * rows * c, s);
  hipLaunchKernelGGL(bn_bwd_apply, dim3(ew_grid((int64_t)rows * (cp / 4))), dim3(256), 0, s, dy, ld_dy, x, ld_x, dx, ld_dx, rows, c, abc, relu_mask);
  TG_CHECK_LAUNCH("bn_bwd_apply");
  return TG_OK;
}

int tg_mobn_finalize_f32(const float* sums, const int32_t* seg_rows, int nseg, int rows, int c, const float* b, float* pop_mean, float decay,
                         int train, float* shift, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(pop_mean && shift && (sums || !train), "mobn_finalize: null buffer");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 0, s);
  hipLaunchKernelGGL(mobn_finalize, dim3((c + 127) / 128), dim3(128), 0, s, sums, st, c, b, pop_mean, decay, train, shift);
  TG_CHECK_LAUNCH("mobn_finalize");
  return TG_OK;
}

// Row-chunk size of the statistics launches: ~2 048 workgroups over all segments for a large tensor (a streaming pass needs
// many waves in flight), never less than 32 rows.  Every workgroup ends with one fp64 atomic per column into replica
// (workgroup % REPL) of the accumulator: <= 64 same-address adds per replica and segment, a tail of a few microseconds.
static int stats_chunk(const SegTable& st) {
  int64_t tot = 0;
  for (int s = 0; s < st.nseg; ++s) tot += st.rows[s];
  int ch = (int)(((tot + 2047) / 2048 + 31) / 32 * 32);
  return ch < 32 ? 32 : ch;
}

static int seg_chunks(const SegTable& st, int chunk) {
  int n = 0;
  for (int s = 0; s < st.nseg; ++s) n += (st.rows[s] + chunk - 1) / chunk;
  return n;
}

int tg_mobn_apply_f32(float* x, int ld, int rows, int c, const int32_t* seg_rows, int nseg, const double* sums, const float* b, float* pop_mean,
                      float decay, int act, float alpha, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(x && pop_mean && c > 0 && c <= 512 && c % 4 == 0 && ld % 4 == 0 && c <= ld, "mobn_apply: c=%d ld=%d unsupported", c, ld);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 8.0 * rows * c, s);
  hipLaunchKernelGGL(mobn_apply, dim3(seg_chunks(st, 32)), dim3(256), 0, s, x, ld, rows, c, st, sums, b, pop_mean, decay, act, alpha);
  TG_CHECK_LAUNCH("mobn_apply");
  return TG_OK;
}

int tg_mobn_apply_pool_f32(float* x, int ld, int n, int h, int w, int c, const int32_t* seg_rows, int nseg, const double* sums, const float* b,
                           float* pop_mean, float decay, int act, float alpha, float* out, int ld_out, const float* mask, int ld_mask, float mscale,
                           void* stream) {
  SegTable st;
  const int rows = n * h * w;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(x && out && pop_mean && n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && c > 0 && c <= 512 && c % 4 == 0 && ld % 4 == 0 && c <= ld &&
             ld_out % 4 == 0 && c <= ld_out && (!mask || (ld_mask % 4 == 0 && c <= ld_mask)), "mobn_apply_pool: c=%d ld=%d ld_out=%d h=%d w=%d unsupported", c, ld,
             ld_out, h, w);
  for (int i = 0; i < nseg; ++i)
    TG_REQUIRE(seg_rows[i] % (h * w) == 0, "mobn_apply_pool: segment %d (%d rows) is not a whole number of %dx%d images", i, seg_rows[i], h, w);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 4.0 * rows * c * 2.25, s);
  hipLaunchKernelGGL(mobn_apply_pool, dim3(n * (h / 2)), dim3(256), 0, s, x, ld, h, w, c, st, sums, b, pop_mean, decay, act, alpha, out, ld_out, mask, ld_mask,
                     mscale);
  TG_CHECK_LAUNCH("mobn_apply_pool");
  return TG_OK;
}

int tg_mobn_bwd_f32(const float* dy, int ld_dy, const float* yact, int ld_y, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg,
                    int act, float alpha, double* sums, int sums_zeroed, float* db, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(dy && yact && dx && sums, "mobn_bwd: null buffer");
  TG_REQUIRE(c > 0 && c <= 512 && c % 4 == 0 && ld_dy % 4 == 0 && ld_y % 4 == 0 && ld_dx % 4 == 0 && c <= ld_dy && c <= ld_y && c <= ld_dx,
             "mobn_bwd: c=%d vs ld unsupported", c);
  hipStream_t s = tg::as_stream(stream);
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * REPL * nseg * c, s);
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(mobn_bwd sums)");
  }
  tg::ProfScope prof(tg::PC_NORM, 0, 20.0 * rows * c, s);
  const int chunk = stats_chunk(st);                             // <= 128 workgroups per segment (atomic tail, see stats_chunk)
  hipLaunchKernelGGL(mobn_bwd_sums, dim3(seg_chunks(st, chunk)), dim3(256), 0, s, dy, ld_dy, yact, ld_y, rows, c, st, act, alpha, sums, chunk);
  TG_CHECK_LAUNCH("mobn_bwd_sums");
  hipLaunchKernelGGL(mobn_bwd_apply, dim3(seg_chunks(st, 32)), dim3(256), 0, s, dy, ld_dy, yact, ld_y, dx, ld_dx, rows, c, st, act, alpha, sums, db);
  TG_CHECK_LAUNCH("mobn_bwd_apply");
  return TG_OK;
}

int tg_mobn_center_f32(const float* t, int ld_t, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg, const double* sums,
                       int n_repl, float* db, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(t && dx && sums && c > 0 && c <= 512 && c % 4 == 0 && ld_t % 4 == 0 && ld_dx % 4 == 0 && c <= ld_t && c <= ld_dx, "mobn_center: c=%d vs ld", c);
  TG_REQUIRE(n_repl == 1 || n_repl == REPL, "mobn_center: n_repl=%d (1 or %d)", n_repl, REPL);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 8.0 * rows * c, s);
  hipLaunchKernelGGL(mobn_center, dim3(seg_chunks(st, 32)), dim3(256), 0, s, t, ld_t, dx, ld_dx, c, st, sums, n_repl, db);
  TG_CHECK_LAUNCH("mobn_center");
  return TG_OK;
}

int tg_maxpool2_bwd_actsum_f32(const float* dout, int ld_do, const float* mask, int ld_mask, float mscale, const float* y, int ld_y, float* t, int ld_t,
                               int n, int h, int w, int c, const int32_t* seg_rows, int nseg, int act, float alpha, double* sums, int sums_zeroed,
                               void* stream) {
  TG_REQUIRE(dout && y && t && sums && n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "maxpool2_bwd_actsum: bad args");
  TG_REQUIRE(c > 0 && c <= 512 && c % 4 == 0 && ld_do % 4 == 0 && ld_y % 4 == 0 && ld_t % 4 == 0 && (!mask || ld_mask % 4 == 0) && c <= ld_do &&
             c <= ld_y && c <= ld_t, "maxpool2_bwd_actsum: c=%d vs ld", c);
  SegTable st;                                           // given in pre-pool pixel rows (the layer's output rows) -> pooled pixels
  int rc = make_segs(st, seg_rows, nseg, n * h * w);
  if (rc != TG_OK) return rc;
  for (int i = 0; i < nseg; ++i) { TG_REQUIRE(st.rows[i] % (h * w) == 0, "maxpool2_bwd_actsum: segment %d is not whole images", i); st.rows[i] /= 4; }
  hipStream_t s = tg::as_stream(stream);
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * REPL * nseg * c, s);
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(maxpool2_bwd_actsum sums)");
  }
  tg::ProfScope prof(tg::PC_ELEMWISE, 0, 4.0 * n * h * w * c * 2.5, s);
  const int chunk = stats_chunk(st);
  hipLaunchKernelGGL(maxpool2_bwd_actsum, dim3(seg_chunks(st, chunk)), dim3(256), 0, s, dout, ld_do, mask, ld_mask, mscale, y, ld_y, t, ld_t, h, w, c, st,
                     chunk, act, alpha, sums);
  TG_CHECK_LAUNCH("maxpool2_bwd_actsum");
  return TG_OK;
}

int tg_actgrad_bias_f32(const float* dy, int ld_dy, const float* yact, int ld_y, float* out, int ld_out, int rows, int c, int act, float alpha,
                        double* sums, int sums_zeroed, float* bias_grad, void* stream) {
  TG_REQUIRE(dy && out && sums && bias_grad && rows > 0 && c > 0, "actgrad_bias: bad args");
  TG_REQUIRE(c <= ld_dy && c <= ld_out && (!yact || c <= ld_y) && ld_out % 4 == 0 && ld_out <= 1024, "actgrad_bias: c=%d ld_out=%d unsupported", c, ld_out);
  hipStream_t s = tg::as_stream(stream);
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * REPL * c, s);
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(actgrad_bias sums)");
  }
  tg::ProfScope prof(tg::PC_ELEMWISE, 0, 4.0 * rows * (ld_out + 2 * c), s);
  int chunk = ((rows + 2047) / 2048 + 31) / 32 * 32;
  if (chunk < 32) chunk = 32;
  hipLaunchKernelGGL(actgrad_bias, dim3((rows + chunk - 1) / chunk), dim3(256), 0, s, dy, ld_dy, yact, ld_y, out, ld_out, rows, c, act, alpha, chunk, sums);
  TG_CHECK_LAUNCH("actgrad_bias");
  hipLaunchKernelGGL(repl_finalize, dim3((c + 255) / 256), dim3(256), 0, s, sums, c, bias_grad);
  TG_CHECK_LAUNCH("repl_finalize");
  return TG_OK;
}

int tg_mobn_bwd_finalize_f32(const float* sums, const int32_t* seg_rows, int nseg, int rows, int c, float* shift, float* db, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(sums && shift && db, "mobn_bwd_finalize: null buffer");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 0, s);
  hipLaunchKernelGGL(mobn_bwd_finalize, dim3((c + 127) / 128), dim3(128), 0, s, sums, st, c, shift, db);
  TG_CHECK_LAUNCH("mobn_bwd_finalize");
  return TG_OK;
}

static int bn_grid(const SegTable& st, int rows, int c, int* chunk, dim3* grid) {
  const int ch = stats_chunk(st);
  int n = 0;
  for (int s = 0; s < st.nseg; ++s) n += (st.rows[s] + ch - 1) / ch;
  *chunk = ch;
  *grid = dim3(n, (c + BN_CW - 1) / BN_CW);
  return TG_OK;
}

int tg_bn_train_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, const int32_t* seg_rows, int nseg, const float* gamma,
                    const float* beta, float eps, float decay, float* moving_mean, float* moving_var, double* sums, int sums_zeroed, float* mean_inv,
                    void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(x && y && gamma && beta && sums && mean_inv, "bn_train: null buffer");
  TG_REQUIRE(c > 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && (c + 3) / 4 * 4 <= ld_x && (c + 3) / 4 * 4 <= ld_y, "bn_train: c=%d vs ld=%d/%d", c, ld_x, ld_y);
  TG_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_train: moving_mean / moving_var must both be given or both be NULL");
  hipStream_t s = tg::as_stream(stream);
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * REPL * 2 * nseg * c, s);
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(bn sums)");
  }
  tg::ProfScope prof(tg::PC_NORM, 0, 12.0 * rows * c, s);
  int chunk; dim3 grid;
  bn_grid(st, rows, c, &chunk, &grid);
  hipLaunchKernelGGL(bn_sums<false>, grid, dim3(256), 0, s, x, ld_x, (const float*)nullptr, 0, c, st, chunk, sums);
  TG_CHECK_LAUNCH("bn_sums");
  hipLaunchKernelGGL(bn_train_apply, grid, dim3(256), 0, s, x, ld_x, y, ld_y, c, st, chunk, sums, gamma, beta, eps, decay, moving_mean, moving_var, mean_inv);
  TG_CHECK_LAUNCH("bn_train_apply");
  return TG_OK;
}

int tg_bn_train_apply_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, const int32_t* seg_rows, int nseg, const float* gamma,
                          const float* beta, float eps, float decay, float* moving_mean, float* moving_var, const double* sums, float* mean_inv,
                          void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(x && y && gamma && beta && sums && mean_inv, "bn_train_apply: null buffer");
  TG_REQUIRE(c > 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && (c + 3) / 4 * 4 <= ld_x && (c + 3) / 4 * 4 <= ld_y, "bn_train_apply: c=%d vs ld=%d/%d", c, ld_x, ld_y);
  TG_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_train_apply: moving_mean / moving_var must both be given or both be NULL");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 8.0 * rows * c, s);
  int chunk; dim3 grid;
  bn_grid(st, rows, c, &chunk, &grid);
  hipLaunchKernelGGL(bn_train_apply, grid, dim3(256), 0, s, x, ld_x, y, ld_y, c, st, chunk, sums, gamma, beta, eps, decay, moving_mean, moving_var, mean_inv);
  TG_CHECK_LAUNCH("bn_train_apply");
  return TG_OK;
}

int tg_bn_moving_update_f32(const double* sums, int rows, int c, const int32_t* seg_rows, int nseg, float decay, float* moving_mean,
                            float* moving_var, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(sums && moving_mean && moving_var, "bn_moving_update: null buffer");
  TG_REQUIRE(c > 0, "bn_moving_update: c=%d", c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 16.0 * c, s);
  hipLaunchKernelGGL(bn_moving_update, dim3((c + 255) / 256), dim3(256), 0, s, sums, c, st, decay, moving_mean, moving_var);
  TG_CHECK_LAUNCH("bn_moving_update");
  return TG_OK;
}

static int bn_train_bwd_impl(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg,
                             const float* gamma, const float* mean_inv, int act, float alpha, double* sums, int sums_zeroed, float* dgamma, float* dbeta,
                             double* dsum, int dsum_zeroed, float* dbias, void* stream);

int tg_bn_train_bwd_f32(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg,
                        const float* gamma, const float* mean_inv, int relu_input, double* sums, int sums_zeroed, float* dgamma, float* dbeta,
                        void* stream) {
  return bn_train_bwd_impl(dy, ld_dy, x, ld_x, dx, ld_dx, rows, c, seg_rows, nseg, gamma, mean_inv, relu_input ? TG_ACT_RELU : TG_ACT_NONE, 0.f, sums,
                           sums_zeroed, dgamma, dbeta, nullptr, 0, nullptr, stream);
}

int tg_bn_train_bwd_act_f32(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows,
                            int nseg, const float* gamma, const float* mean_inv, int act, float alpha, double* sums, int sums_zeroed, float* dgamma,
                            float* dbeta, double* dsum, int dsum_zeroed, float* dbias, void* stream) {
  TG_REQUIRE(act == TG_ACT_NONE || act == TG_ACT_RELU || act == TG_ACT_LRELU, "bn_train_bwd_act: activation %d has no derivative from its output here", act);
  TG_REQUIRE((dsum == nullptr) == (dbias == nullptr), "bn_train_bwd_act: dsum / dbias must both be given or both be NULL");
  TG_REQUIRE(c > 0, "bn_train_bwd_act: c=%d", c);
  const int ncol = c < BN_CW ? c : BN_CW;
  TG_REQUIRE(dsum == nullptr || (c % 4 == 0 && 256 % ((ncol + 3) / 4) == 0 && (c <= BN_CW || c % BN_CW == 0)),
             "bn_train_bwd_act: the bias-gradient sums need 4 | c and column groups that divide 256 (c=%d)", c);
  return bn_train_bwd_impl(dy, ld_dy, x, ld_x, dx, ld_dx, rows, c, seg_rows, nseg, gamma, mean_inv, act, alpha, sums, sums_zeroed, dgamma, dbeta, dsum,
                           dsum_zeroed, dbias, stream);
}

static int bn_train_bwd_impl(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg,
                             const float* gamma, const float* mean_inv, int act, float alpha, double* sums, int sums_zeroed, float* dgamma, float* dbeta,
                             double* dsum, int dsum_zeroed, float* dbias, void* stream) {
  SegTable st;
  int rc = make_segs(st, seg_rows, nseg, rows);
  if (rc != TG_OK) return rc;
  TG_REQUIRE(dy && x && dx && gamma && mean_inv && sums, "bn_train_bwd: null buffer");
  TG_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_train_bwd: dgamma / dbeta must both be given or both be NULL");
  const int cp = (c + 3) / 4 * 4;
  TG_REQUIRE(c > 0 && ld_dy % 4 == 0 && ld_x % 4 == 0 && ld_dx % 4 == 0 && cp <= ld_dy && cp <= ld_x && cp <= ld_dx, "bn_train_bwd: c=%d vs ld", c);
  hipStream_t s = tg::as_stream(stream);
  const bool sums_given = sums_zeroed == 2;               // the launch that produced dy took the statistics in its epilogue (tg_igemm_bnbwdstat_*)
  if (!sums_zeroed) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * REPL * 2 * nseg * c, s);
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(bn bwd sums)");
  }
  if (dsum && !dsum_zeroed) {
    hipError_t e = hipMemsetAsync(dsum, 0, sizeof(double) * REPL * c, s);
    if (e != hipSuccess) return tg::hip_fail(e, "hipMemsetAsync(bn bwd bias sums)");
  }
  tg::ProfScope prof(tg::PC_NORM, 0, 20.0 * rows * c, s);
  int chunk; dim3 grid;
  bn_grid(st, rows, c, &chunk, &grid);
  if (!sums_given) {
    hipLaunchKernelGGL(bn_sums<true>, grid, dim3(256), 0, s, dy, ld_dy, x, ld_x, c, st, chunk, sums);
    TG_CHECK_LAUNCH("bn_sums<bwd>");
  }
  hipLaunchKernelGGL(bn_train_bwd_apply, grid, dim3(256), 0, s, dy, ld_dy, x, ld_x, dx, ld_dx, c, st, chunk, sums, gamma, mean_inv, act, alpha, dgamma, dbeta,
                     dsum);
  TG_CHECK_LAUNCH("bn_train_bwd_apply");
  if (dsum) {
    hipLaunchKernelGGL(repl_finalize, dim3((c + 255) / 256), dim3(256), 0, s, dsum, c, dbias);
    TG_CHECK_LAUNCH("repl_finalize");
  }
  return TG_OK;
}

int tg_bn_finalize_f32(const float* s1, const float* s2, int rows, int c, const float* gamma, const float* beta, float eps, float* scale,
                       float* shift, float* mean_inv, float* moving_mean, float* moving_var, float decay, int bessel, void* stream) {
  TG_REQUIRE(s1 && s2 && gamma && beta && scale && shift && mean_inv, "bn_finalize: null buffer");
  TG_REQUIRE(rows > 0 && c > 0, "bn_finalize: rows=%d c=%d (statistics of an empty batch are undefined)", rows, c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 0, s);
  hipLaunchKernelGGL(bn_finalize, dim3((c + 127) / 128), dim3(128), 0, s, s1, s2, rows, c, gamma, beta, eps, scale, shift, mean_inv, moving_mean,
                     moving_var, decay, bessel);
  TG_CHECK_LAUNCH("bn_finalize");
  return TG_OK;
}

int tg_bn_eval_finalize_f32(int c, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                            float* scale, float* shift, void* stream) {
  TG_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && c > 0, "bn_eval_finalize: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 0, s);
  hipLaunchKernelGGL(bn_eval_finalize, dim3((c + 127) / 128), dim3(128), 0, s, c, gamma, beta, moving_mean, moving_var, eps, scale, shift);
  TG_CHECK_LAUNCH("bn_eval_finalize");
  return TG_OK;
}

int tg_bn_bwd_finalize_f32(const float* s_dy, const float* s_dyx, int rows, int c, const float* gamma, const float* mean_inv, float* abc,
                           float* dgamma, float* dbeta, void* stream) {
  TG_REQUIRE(s_dy && s_dyx && gamma && mean_inv && abc && dgamma && dbeta, "bn_bwd_finalize: null buffer");
  TG_REQUIRE(rows > 0 && c > 0, "bn_bwd_finalize: rows=%d c=%d", rows, c);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_NORM, 0, 0, s);
  hipLaunchKernelGGL(bn_bwd_finalize, dim3((c + 127) / 128), dim3(128), 0, s, s_dy, s_dyx, rows, c, gamma, mean_inv, abc, dgamma, dbeta);
  TG_CHECK_LAUNCH("bn_bwd_finalize");
  return TG_OK;
}

}  // extern "C"
