// Minibatch discrimination (NN_Base._minibatch_discrimination, Model/modle_base.py:110-128; the MINIBATCH_DIS branch of the SVHN
// discriminator, Model/Good_GAN.py:159-162): with A = reshape(x @ W, [N, K, D]),
//     f[i,k] = sum_j exp(-sum_d |A[i,k,d] - A[j,k,d]|) + b[k]            (the j = i term contributes 1)
// and the layer's output is concat([x, f], 1).  N <= 256 rows, K = 100 kernels of D = 5: 3e7 exponentials — one thread per (i, k)
// walks j; A (N*K*D floats, 0.5 MB) stays in L2.  The product x @ W itself is a dense tg_igemm_f32 launch.
#include "tg_common.h"
#include "tg_device.h"

namespace {

constexpr int MAX_DIM = 8;

__global__ void __launch_bounds__(256) mbd_fwd(const float* __restrict__ act, int ld_a, const float* __restrict__ x, int ld_x, int c,
                                               const float* __restrict__ b, float* __restrict__ out, int ld_out, int n, int nk, int dim) {
  const int64_t total = (int64_t)n * ld_out;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / ld_out), col = (int)(idx - (int64_t)i * ld_out);
    float v = 0.f;
    if (col < c) {
      v = x[(int64_t)i * ld_x + col];
    } else if (col < c + nk) {
      const int k = col - c;
      float ai[MAX_DIM];
      for (int d = 0; d < dim; ++d) ai[d] = act[(int64_t)i * ld_a + k * dim + d];
      for (int j = 0; j < n; ++j) {
        float s = 0.f;
        for (int d = 0; d < dim; ++d) s += fabsf(ai[d] - act[(int64_t)j * ld_a + k * dim + d]);
        v += expf(-s);
      }
      v += b[k];
    }
    out[idx] = v;
  }
}

// g = d loss / d f  ([n][nk], row stride ld_g).  dact[i,k,d] = -sum_j (g[i,k] + g[j,k]) e_ijk sign(a_ikd - a_jkd); pads of dact zeroed.
__global__ void __launch_bounds__(256) mbd_bwd(const float* __restrict__ act, int ld_a, const float* __restrict__ g, int ld_g, float* __restrict__ dact,
                                               int ld_da, int n, int nk, int dim) {
  const int64_t total = (int64_t)n * nk;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / nk), k = (int)(idx - (int64_t)i * nk);
    float ai[MAX_DIM], acc[MAX_DIM];
    for (int d = 0; d < dim; ++d) { ai[d] = act[(int64_t)i * ld_a + k * dim + d]; acc[d] = 0.f; }
    const float gi = g[(int64_t)i * ld_g + k];
    for (int j = 0; j < n; ++j) {
      float s = 0.f, df[MAX_DIM];
      for (int d = 0; d < dim; ++d) { df[d] = ai[d] - act[(int64_t)j * ld_a + k * dim + d]; s += fabsf(df[d]); }
      const float e = expf(-s) * (gi + g[(int64_t)j * ld_g + k]);
      for (int d = 0; d < dim; ++d) acc[d] -= e * (df[d] > 0.f ? 1.f : (df[d] < 0.f ? -1.f : 0.f));
    }
    for (int d = 0; d < dim; ++d) dact[(int64_t)i * ld_da + k * dim + d] = acc[d];
  }
  const int pad = ld_da - nk * dim;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)n * pad; idx += (int64_t)gridDim.x * 256)
    dact[(idx / pad) * ld_da + nk * dim + idx % pad] = 0.f;
}

__global__ void __launch_bounds__(256) mbd_bias_grad(const float* __restrict__ g, int ld_g, float* __restrict__ db, int n, int nk) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < nk; k += gridDim.x * 256) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += g[(int64_t)i * ld_g + k];
    db[k] = s;
  }
}

int grid_for(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" {

int tg_minibatch_disc_fwd_f32(const float* act, int ld_a, const float* x, int ld_x, int c, const float* b, float* out, int ld_out, int n, int nk,
                              int dim, void* stream) {
  TG_REQUIRE(act && x && b && out, "minibatch_disc_fwd: null buffer");
  TG_REQUIRE(n > 0 && nk > 0 && dim > 0 && dim <= MAX_DIM && c >= 0 && c <= ld_x && nk * dim <= ld_a && c + nk <= ld_out,
             "minibatch_disc_fwd: n=%d kernels=%d dim=%d c=%d ld_a=%d ld_x=%d ld_out=%d", n, nk, dim, c, ld_a, ld_x, ld_out);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_ELEMWISE, 0, 4.0 * n * (ld_out + c + (double)nk * dim), s);
  hipLaunchKernelGGL(mbd_fwd, dim3(grid_for((int64_t)n * ld_out)), dim3(256), 0, s, act, ld_a, x, ld_x, c, b, out, ld_out, n, nk, dim);
  TG_CHECK_LAUNCH("mbd_fwd");
  return TG_OK;
}

int tg_minibatch_disc_bwd_f32(const float* act, int ld_a, const float* df, int ld_df, float* dact, int ld_da, float* db, int n, int nk, int dim,
                              void* stream) {
  TG_REQUIRE(act && df && dact, "minibatch_disc_bwd: null buffer");
  TG_REQUIRE(n > 0 && nk > 0 && dim > 0 && dim <= MAX_DIM && nk * dim <= ld_a && nk * dim <= ld_da && nk <= ld_df,
             "minibatch_disc_bwd: n=%d kernels=%d dim=%d ld_a=%d ld_da=%d ld_df=%d", n, nk, dim, ld_a, ld_da, ld_df);
  hipStream_t s = tg::as_stream(stream);
  tg::ProfScope prof(tg::PC_ELEMWISE, 0, 4.0 * n * (2.0 * nk * dim + nk), s);
  hipLaunchKernelGGL(mbd_bwd, dim3(grid_for((int64_t)n * nk)), dim3(256), 0, s, act, ld_a, df, ld_df, dact, ld_da, n, nk, dim);
  TG_CHECK_LAUNCH("mbd_bwd");
  if (db) {
    hipLaunchKernelGGL(mbd_bias_grad, dim3(grid_for(nk)), dim3(256), 0, s, df, ld_df, db, n, nk);
    TG_CHECK_LAUNCH("mbd_bias_grad");
  }
  return TG_OK;
}

}  // extern "C"
