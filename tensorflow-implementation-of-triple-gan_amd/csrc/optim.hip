// Multi-tensor Adam (TF form) and EMA over a network's FLAT parameter buffer: one launch per network,
// 28 B/param (read p,g,m,v; write p,m,v) — HBM-bound, 16-B lanes, grid-stride.
// Hyper-parameters that change between iterations (learning rate, step count) live in device memory so that
// a captured hipGraph replays with fresh values.
//   tf.train.AdamOptimizer (Training/train_base.py:91-97) in its ApplyAdam functor form: alpha = lr*sqrt(1-b2^t)/(1-b1^t);
//   m += (g-m)(1-b1); v += (g^2-v)(1-b2); p -= m*alpha/(sqrt(v)+eps)              [UNVERIFIED-TF]
//   tf.train.ExponentialMovingAverage(0.9999).apply (Training/Train_goodGAN.py:101-103): s -= (1-d)(s-p)
#include "tg_common.h"

namespace {

__global__ void step_inc(int* t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[0] += 1; }

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, const float* __restrict__ lr_ptr, float beta1, float beta2, float eps,
                                                   const int* __restrict__ t_ptr, float grad_scale) {
  const int t = t_ptr[0];
  const float lr_t = (float)((double)lr_ptr[0] * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t)));
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * grad_scale;
      mp[k] = mp[k] + (gg - mp[k]) * omb1;
      vp[k] = vp[k] + (gg * gg - vp[k]) * omb2;
      pp[k] = pp[k] - mp[k] * lr_t / (sqrtf(vp[k]) + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gg = g[i] * grad_scale;
    const float mm = m[i] + (gg - m[i]) * omb1, vv = v[i] + (gg * gg - v[i]) * omb2;
    m[i] = mm; v[i] = vv;
    p[i] = p[i] - mm * lr_t / (sqrtf(vv) + eps);
  }
}

__global__ void __launch_bounds__(256) ema_kernel(float* __restrict__ s, const float* __restrict__ p, int64_t n, float one_minus_decay) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s[i] = s[i] - one_minus_decay * (s[i] - p[i]);
}

int ew_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" {

int tg_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2, float eps, int* step_dev,
                float grad_scale, void* stream) {
  TG_REQUIRE(p && g && m && v && lr_dev && step_dev && n > 0, "adam: bad args");
  TG_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "adam: buffers must be 16-B aligned");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_OPTIM, 0, 28.0 * n, s);
  hipLaunchKernelGGL(step_inc, dim3(1), dim3(64), 0, s, step_dev);
  TG_CHECK_LAUNCH("step_inc");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, s, p, g, m, v, n, lr_dev, beta1, beta2, eps, step_dev, grad_scale);
  TG_CHECK_LAUNCH("adam_kernel");
  return TG_OK;
}

int tg_ema_f32(float* shadow, const float* p, int64_t n, float decay, void* stream) {
  TG_REQUIRE(shadow && p && n > 0, "ema: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_OPTIM, 0, 12.0 * n, s);
  hipLaunchKernelGGL(ema_kernel, dim3(ew_grid(n)), dim3(256), 0, s, shadow, p, n, 1.f - decay);
  TG_CHECK_LAUNCH("ema_kernel");
  return TG_OK;
}

}  // extern "C"
