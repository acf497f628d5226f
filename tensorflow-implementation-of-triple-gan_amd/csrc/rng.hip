// Counter-based RNG (Philox4x32-10) for the stochastic ops of the step: dropout keep-masks
// (tf.layers.dropout, Model/modle_base.py:190-191), Gaussian input noise (tf.random_normal,
// Model/modle_base.py:193-202), the latent z ~ U(-1,1) and y ~ onehot(U{0..9}) draws of
// Training/Train_goodGAN.py:234-239.  The (seed, step) pair lives in DEVICE memory: a captured hipGraph
// draws fresh numbers on every replay after tg_rng_advance.  counter = (index/4, stream_id, step), key = seed.
#include "tg_common.h"

namespace {

struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

// mode 0: uniform [lo,hi); 1: bernoulli keep-mask (1 with prob a); 2: normal(0, a); 3: one-hot of k=a classes per row
__global__ void __launch_bounds__(256) rng_kernel(float* __restrict__ out, int64_t n, int mode, float a, float b, const uint64_t* __restrict__ state,
                                                  uint32_t stream_id) {
  const uint64_t seed = state[0], step = state[1];
  const int64_t n4 = (n + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const u4 r = philox((uint32_t)i, (uint32_t)(i >> 32), stream_id, (uint32_t)step, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
    float v[4];
    if (mode == 2) {
      const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
      sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
      v[0] = a * r0 * c0; v[1] = a * r0 * s0; v[2] = a * r1 * c1; v[3] = a * r1 * s1;
    } else {
      const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = mode == 0 ? a + (b - a) * u[k] : (u[k] < a ? 1.f : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i * 4 + k < n) out[i * 4 + k] = v[k];
  }
}

__global__ void rng_onehot_kernel(float* __restrict__ out, int rows, int k, const uint64_t* __restrict__ state, uint32_t stream_id) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const uint64_t seed = state[0], step = state[1];
  const u4 q = philox((uint32_t)r, 0u, stream_id, (uint32_t)step, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
  const int cls = (int)(((uint64_t)q.x * (uint64_t)k) >> 32);
  for (int j = 0; j < k; ++j) out[r * k + j] = j == cls ? 1.f : 0.f;
}

constexpr int MAX_RNG_JOBS = 16;
struct RngJobs { tg_rng_job j[MAX_RNG_JOBS]; };

// blockIdx.y = job; per job exactly the arithmetic of rng_kernel / rng_onehot_kernel
__global__ void __launch_bounds__(256) rng_multi_kernel(RngJobs js, const uint64_t* __restrict__ state) {
  const tg_rng_job& J = js.j[blockIdx.y];
  const uint64_t seed = state[0], step = state[1];
  if (J.mode == 3) {
    const int k = (int)J.a;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < J.n; r += (int64_t)gridDim.x * blockDim.x) {
      const u4 q = philox((uint32_t)r, 0u, J.stream_id, (uint32_t)step, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
      const int cls = (int)(((uint64_t)q.x * (uint64_t)k) >> 32);
      for (int c = 0; c < k; ++c) J.out[r * k + c] = c == cls ? 1.f : 0.f;
    }
    return;
  }
  const int64_t n4 = (J.n + 3) / 4;
  const float a = J.a, b = J.b;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const u4 r = philox((uint32_t)i, (uint32_t)(i >> 32), J.stream_id, (uint32_t)step, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
    float v[4];
    if (J.mode == 2) {
      const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
      sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
      v[0] = a * r0 * c0; v[1] = a * r0 * s0; v[2] = a * r1 * c1; v[3] = a * r1 * s1;
    } else {
      const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = J.mode == 0 ? a + (b - a) * u[k] : (u[k] < a ? 1.f : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i * 4 + k < J.n) J.out[i * 4 + k] = v[k];
  }
}

__global__ void rng_advance_kernel(uint64_t* state) { if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += 1; }

int ew_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch(float* out, int64_t n, int mode, float a, float b, const uint64_t* state, uint32_t stream_id, void* stream) {
  TG_REQUIRE(out && state && n > 0, "rng: bad args");
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_ELEMWISE, 0, 4.0 * n, s);
  hipLaunchKernelGGL(rng_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, s, out, n, mode, a, b, state, stream_id);
  TG_CHECK_LAUNCH("rng_kernel");
  return TG_OK;
}

}  // namespace

extern "C" {

int tg_rng_uniform_f32(float* out, int64_t n, float lo, float hi, const uint64_t* state, uint32_t stream_id, void* stream) {
  return launch(out, n, 0, lo, hi, state, stream_id, stream);
}
int tg_rng_keep_mask_f32(float* out, int64_t n, float keep_prob, const uint64_t* state, uint32_t stream_id, void* stream) {
  return launch(out, n, 1, keep_prob, 0.f, state, stream_id, stream);
}
int tg_rng_normal_f32(float* out, int64_t n, float stddev, const uint64_t* state, uint32_t stream_id, void* stream) {
  return launch(out, n, 2, stddev, 0.f, state, stream_id, stream);
}
int tg_rng_onehot_f32(float* out, int rows, int k, const uint64_t* state, uint32_t stream_id, void* stream) {
  TG_REQUIRE(out && state && rows > 0 && k > 0, "rng_onehot: bad args");
  hipStream_t s = tg::as_stream(stream);
  hipLaunchKernelGGL(rng_onehot_kernel, dim3((rows + 127) / 128), dim3(128), 0, s, out, rows, k, state, stream_id);
  TG_CHECK_LAUNCH("rng_onehot_kernel");
  return TG_OK;
}
int tg_rng_multi_f32(const tg_rng_job* jobs, int n_jobs, const uint64_t* state, void* stream) {
  TG_REQUIRE(jobs && state && n_jobs >= 1 && n_jobs <= MAX_RNG_JOBS, "rng_multi: n_jobs=%d out of range", n_jobs);
  RngJobs js;
  int64_t mx = 0;
  double bytes = 0;
  for (int i = 0; i < n_jobs; ++i) {
    TG_REQUIRE(jobs[i].out && jobs[i].n > 0 && jobs[i].mode >= 0 && jobs[i].mode <= 3, "rng_multi: job %d bad", i);
    js.j[i] = jobs[i];
    const int64_t work = jobs[i].mode == 3 ? jobs[i].n : (jobs[i].n + 3) / 4;
    mx = work > mx ? work : mx;
    bytes += 4.0 * jobs[i].n * (jobs[i].mode == 3 ? jobs[i].a : 1.f);
  }
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_ELEMWISE, 0, bytes, s);
  hipLaunchKernelGGL(rng_multi_kernel, dim3(ew_grid(mx), n_jobs), dim3(256), 0, s, js, state);
  TG_CHECK_LAUNCH("rng_multi_kernel");
  return TG_OK;
}

int tg_rng_advance(uint64_t* state, void* stream) {
  TG_REQUIRE(state, "rng_advance: null state");
  hipStream_t s = tg::as_stream(stream);
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, s, state);
  TG_CHECK_LAUNCH("rng_advance_kernel");
  return TG_OK;
}

}  // extern "C"
