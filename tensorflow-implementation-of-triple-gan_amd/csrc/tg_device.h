// Device-side helpers shared by the elementwise / norm kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tg_kernels.h"

namespace tgd {

__device__ __forceinline__ float act(float v, int a, float alpha) {
  switch (a) {
    case TG_ACT_LRELU: return v > 0.f ? v : alpha * v;       // relu(x) - alpha*relu(-x), Good_GAN_cifar10.py:26-27
    case TG_ACT_RELU: return v > 0.f ? v : 0.f;
    case TG_ACT_TANH: return tanhf(v);
    case TG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case TG_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
    default: return v;
  }
}

// derivative expressed through the activation OUTPUT y (what the forward pass keeps)
__device__ __forceinline__ float act_grad(float y, int a, float alpha) {
  switch (a) {
    case TG_ACT_LRELU: return y > 0.f ? 1.f : alpha;
    case TG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case TG_ACT_TANH: return 1.f - y * y;
    case TG_ACT_SIGMOID: return y * (1.f - y);
    case TG_ACT_SOFTPLUS: return 1.f - expf(-y);             // y = log(1+e^x) -> sigmoid(x) = 1 - e^-y
    default: return 1.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace tgd
