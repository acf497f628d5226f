// Loss heads of Train_base._loss_GAN (Training/train_base.py:113-154) fused per network into ONE
// single-workgroup launch each: value + gradient wrt the logits.  These are latency-bound (<= 250 rows of
// <= 10 logits); reductions are wavefront shuffles + one LDS hop.  Also the feature-matching and pull-away
// terms of train_base.py:172-182,202-207 (unit parity; not on the Train_goodGAN.py path) and the
// streaming accuracy counter of Train_goodGAN.py:428-447.
#include "tg_common.h"
#include "tg_device.h"

namespace {

constexpr int KC = 10;   // NUM_CLASSES of every config of the reference

__device__ float block_sum(float v, float* red) {   // 256 threads = 4 waves
  v = tgd::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float bce(float z, float t) { return fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigm(float z) { return 1.f / (1.f + expf(-z)); }

// rows ordered [real | fake | unl]; d_loss = BCE(real,1) + .5 BCE(fake,0) + .5 BCE(unl,0)
__global__ void __launch_bounds__(256) d_loss_kernel(const float* __restrict__ z, int ld, int n_real, int n_fake, int n_unl, float* __restrict__ dz, int ld_d,
                                                     float* __restrict__ loss) {
  __shared__ float red[4];
  const int n = n_real + n_fake + n_unl;
  float acc = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float v = z[(int64_t)r * ld];
    float t, w;
    if (r < n_real) { t = 1.f; w = 1.f / n_real; }
    else if (r < n_real + n_fake) { t = 0.f; w = 0.5f / n_fake; }
    else { t = 0.f; w = 0.5f / n_unl; }
    acc += w * bce(v, t);
    float* o = dz + (int64_t)r * ld_d;
    o[0] = w * (sigm(v) - t);
    for (int k = 1; k < ld_d; ++k) o[k] = 0.f;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
}

// g_loss = 0.5 * BCE(D_fake, 1)
__global__ void __launch_bounds__(256) g_loss_kernel(const float* __restrict__ z, int ld, int n, float* __restrict__ dz, int ld_d, float* __restrict__ loss) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float v = z[(int64_t)r * ld];
    acc += 0.5f / n * bce(v, 1.f);
    float* o = dz + (int64_t)r * ld_d;
    o[0] = 0.5f / n * (sigm(v) - 1.f);
    for (int k = 1; k < ld_d; ++k) o[k] = 0.f;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
}

__device__ __forceinline__ void softmax10(const float* l, float* p, float* lse) {
  float m = l[0];
#pragma unroll
  for (int k = 1; k < KC; ++k) m = fmaxf(m, l[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KC; ++k) { p[k] = expf(l[k] - m); s += p[k]; }
  const float inv = 1.f / s;
#pragma unroll
  for (int k = 0; k < KC; ++k) p[k] *= inv;
  *lse = m + logf(s);
}

// rows ordered [real | unl | unl_rep (n_rep = n_unl or 0) | fake]
// c_loss = 0.005*c_unl + CE(real) + 1e-6*H(unl) + 1e-3*Bal(unl) + lam1*CE(fake) + lam2*MSE(unl,rep)
__global__ void __launch_bounds__(256) c_loss_kernel(const float* __restrict__ cl, int ld, int n_real, int n_unl, int n_rep, int n_fake,
                                                     const float* __restrict__ y_real, const float* __restrict__ y_fake, const float* __restrict__ d_unl,
                                                     int ld_dunl, const float* __restrict__ lam, float* __restrict__ dl, int ld_d, float* __restrict__ loss) {
  __shared__ float red[4];
  __shared__ float q[KC];
  const float lam1 = lam[0], lam2 = n_rep > 0 ? lam[1] : 0.f;
  const int o_unl = n_real, o_rep = n_real + n_unl, o_fake = o_rep + n_rep;
  float l[KC], p[KC], lse;
  // pass A: balance-entropy class marginals q_k = mean_n softmax(C_unl)_k
  float qa[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) qa[k] = 0.f;
  for (int r = threadIdx.x; r < n_unl; r += 256) {
#pragma unroll
    for (int k = 0; k < KC; ++k) l[k] = cl[(int64_t)(o_unl + r) * ld + k];
    softmax10(l, p, &lse);
#pragma unroll
    for (int k = 0; k < KC; ++k) qa[k] += p[k];
  }
  for (int k = 0; k < KC; ++k) {
    const float s = block_sum(qa[k], red);
    if (threadIdx.x == 0) q[k] = s / n_unl;
  }
  __syncthreads();
  float acc = 0.f;
  if (threadIdx.x == 0) {
    float bal = 0.f;
    for (int k = 0; k < KC; ++k) bal -= logf(q[k] + 1e-12f) / KC;
    acc += 1e-3f * bal;
  }
  // labelled rows (real, then fake)
  for (int pass = 0; pass < 2; ++pass) {
    const int n = pass == 0 ? n_real : n_fake, off = pass == 0 ? 0 : o_fake;
    const float* y = pass == 0 ? y_real : y_fake;
    const float w = pass == 0 ? 1.f : lam1;
    for (int r = threadIdx.x; r < n; r += 256) {
#pragma unroll
      for (int k = 0; k < KC; ++k) l[k] = cl[(int64_t)(off + r) * ld + k];
      softmax10(l, p, &lse);
      float ysum = 0.f, yl = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) { const float t = y[r * KC + k]; ysum += t; yl += t * l[k]; }
      acc += w * (lse * ysum - yl) / n;
      float* o = dl + (int64_t)(off + r) * ld_d;
#pragma unroll
      for (int k = 0; k < KC; ++k) o[k] = w * (p[k] * ysum - y[r * KC + k]) / n;
      for (int k = KC; k < ld_d; ++k) o[k] = 0.f;
    }
  }
  // unlabelled rows
  for (int r = threadIdx.x; r < n_unl; r += 256) {
#pragma unroll
    for (int k = 0; k < KC; ++k) l[k] = cl[(int64_t)(o_unl + r) * ld + k];
    softmax10(l, p, &lse);
    int j = 0;
    float pm = p[0], pl = 0.f, pdq = 0.f;
    float dq[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (p[k] > pm) { pm = p[k]; j = k; }
      pl += p[k] * l[k];
      dq[k] = -1.f / (KC * (q[k] + 1e-12f)) / n_unl;
      pdq += p[k] * dq[k];
    }
    const float rr = bce(d_unl[(int64_t)r * ld_dunl], 1.f);
    acc += 0.005f * pm * rr / n_unl + 1e-6f * (lse - pl) / n_unl;
    float* o = dl + (int64_t)(o_unl + r) * ld_d;
    float* orep = dl + (int64_t)(o_rep + r) * ld_d;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      float g = 0.005f * rr * pm * ((k == j ? 1.f : 0.f) - p[k]) / n_unl;     // C fools D (train_base.py:133-137)
      g += 1e-6f * (p[k] - p[k] * (1.f + l[k] - pl)) / n_unl;                  // entropy
      g += 1e-3f * p[k] * (dq[k] - pdq);                                       // balance entropy
      if (n_rep > 0) {
        const float d = cl[(int64_t)(o_rep + r) * ld + k] - l[k];
        const float gm = lam2 * 2.f * d / (n_unl * KC);
        acc += lam2 * d * d / (n_unl * KC);
        g -= gm;
        orep[k] = gm;
      }
      o[k] = g;
    }
    for (int k = KC; k < ld_d; ++k) { o[k] = 0.f; if (n_rep > 0) orep[k] = 0.f; }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
}

// fm = mean_c | mean_n f_fake - mean_n f_unl |   (train_base.py:172); one thread per feature column
__global__ void __launch_bounds__(256) feature_match_kernel(const float* __restrict__ ff, int n_f, const float* __restrict__ fu, int n_u, int c,
                                                            float* __restrict__ dff, float* __restrict__ dfu, float* __restrict__ loss) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int k = threadIdx.x; k < c; k += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < n_f; ++r) a += ff[r * c + k];
    for (int r = 0; r < n_u; ++r) b += fu[r * c + k];
    const float d = a / n_f - b / n_u;
    acc += fabsf(d) / c;
    const float s = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / c;
    for (int r = 0; r < n_f; ++r) dff[r * c + k] = s / n_f;
    for (int r = 0; r < n_u; ++r) dfu[r * c + k] = -s / n_u;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
}

// pull-away term over N feature rows of width c (N <= 256, c <= 256); masked (train_base.py:175-181) or not (:204-207).
// scratch: N*c normalised rows + N*N cosines + N norms.
__global__ void __launch_bounds__(256) pull_away_kernel(const float* __restrict__ f, int n, int c, int masked, float* __restrict__ scratch,
                                                        float* __restrict__ df, float* __restrict__ loss) {
  __shared__ float red[4];
  float* fn = scratch;
  float* cs = scratch + n * c;
  float* nr = cs + n * n;
  for (int r = threadIdx.x; r < n; r += 256) {
    float ss = 0.f;
    for (int k = 0; k < c; ++k) ss += f[r * c + k] * f[r * c + k];
    const float nn = sqrtf(ss);
    nr[r] = nn;
    for (int k = 0; k < c; ++k) fn[r * c + k] = f[r * c + k] / nn;
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = threadIdx.x; i < n * n; i += 256) {
    const int a = i / n, b = i % n;
    float d = 0.f;
    for (int k = 0; k < c; ++k) d += fn[a * c + k] * fn[b * c + k];
    if (masked) {
      const float mm = a == b ? 0.f : 1.f;
      acc += 0.8f * d * d * mm / (n * (n - 1));
      cs[i] = 0.8f * 2.f * d * mm / (n * (n - 1));      // d loss / d cos
    } else {
      acc += 0.8f * d / (n * n);
      cs[i] = 0.8f / (n * n);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n * c; i += 256) {
    const int r = i / c, k = i % c;
    float g = 0.f;
    for (int b = 0; b < n; ++b) g += (cs[r * n + b] + cs[b * n + r]) * fn[b * c + k];
    df[i] = g;                                            // d loss / d fn (projected below)
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += 256) {
    float dot = 0.f;
    for (int k = 0; k < c; ++k) dot += df[r * c + k] * fn[r * c + k];
    for (int k = 0; k < c; ++k) df[r * c + k] = (df[r * c + k] - fn[r * c + k] * dot) / nr[r];
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
}

// counters[0] += #(argmax logits == argmax labels), counters[1] += n     (tf.metrics.accuracy, Train_goodGAN.py:428-447)
__global__ void __launch_bounds__(256) accuracy_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ labels, int n, int k,
                                                       float* __restrict__ counters) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
    int a = 0, b = 0;
    for (int j = 1; j < k; ++j) {
      if (logits[(int64_t)r * ld + j] > logits[(int64_t)r * ld + a]) a = j;
      if (labels[r * k + j] > labels[r * k + b]) b = j;
    }
    acc += a == b ? 1.f : 0.f;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) { counters[0] += acc; counters[1] += (float)n; }
}

}  // namespace

#define LOSS_LAUNCH(kern, ...)                                               \
  hipStream_t s__ = tg::as_stream(stream);                                   \
  tg::ProfScope prof__(tg::PC_LOSS, 0, 0, s__);                              \
  hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, s__, __VA_ARGS__);         \
  TG_CHECK_LAUNCH(#kern);                                                    \
  return TG_OK;

extern "C" {

int tg_d_loss_f32(const float* logits, int ld, int n_real, int n_fake, int n_unl, float* dlogits, int ld_d, float* loss, void* stream) {
  TG_REQUIRE(logits && dlogits && loss && n_real > 0 && n_fake > 0 && n_unl > 0 && ld >= 1 && ld_d >= 1, "d_loss: bad args");
  LOSS_LAUNCH(d_loss_kernel, logits, ld, n_real, n_fake, n_unl, dlogits, ld_d, loss)
}

int tg_g_loss_f32(const float* logits, int ld, int n, float* dlogits, int ld_d, float* loss, void* stream) {
  TG_REQUIRE(logits && dlogits && loss && n > 0 && ld >= 1 && ld_d >= 1, "g_loss: bad args");
  LOSS_LAUNCH(g_loss_kernel, logits, ld, n, dlogits, ld_d, loss)
}

int tg_c_loss_f32(const float* c_logits, int ld, int n_real, int n_unl, int n_rep, int n_fake, const float* y_real, const float* y_fake,
                  const float* d_unl_logits, int ld_dunl, const float* lambdas, float* dlogits, int ld_d, float* loss, void* stream) {
  TG_REQUIRE(c_logits && y_real && y_fake && d_unl_logits && lambdas && dlogits && loss, "c_loss: null buffer");
  TG_REQUIRE(n_real > 0 && n_unl > 0 && n_fake > 0 && (n_rep == 0 || n_rep == n_unl) && ld >= KC && ld_d >= KC, "c_loss: bad sizes");
  LOSS_LAUNCH(c_loss_kernel, c_logits, ld, n_real, n_unl, n_rep, n_fake, y_real, y_fake, d_unl_logits, ld_dunl, lambdas, dlogits, ld_d, loss)
}

int tg_feature_match_f32(const float* f_fake, int n_fake, const float* f_unl, int n_unl, int c, float* df_fake, float* df_unl, float* loss,
                         void* stream) {
  TG_REQUIRE(f_fake && f_unl && df_fake && df_unl && loss && n_fake > 0 && n_unl > 0 && c > 0, "feature_match: bad args");
  LOSS_LAUNCH(feature_match_kernel, f_fake, n_fake, f_unl, n_unl, c, df_fake, df_unl, loss)
}

/* scratch: n*c + n*n + n floats */
int tg_pull_away_f32(const float* f, int n, int c, int masked, float* scratch, float* df, float* loss, void* stream) {
  TG_REQUIRE(f && scratch && df && loss && n > 1 && c > 0, "pull_away: bad args");
  LOSS_LAUNCH(pull_away_kernel, f, n, c, masked, scratch, df, loss)
}

int tg_accuracy_count_f32(const float* logits, int ld, const float* labels, int n, int k, float* counters, void* stream) {
  TG_REQUIRE(logits && labels && counters && n > 0 && k > 0 && k <= ld, "accuracy_count: bad args");
  LOSS_LAUNCH(accuracy_kernel, logits, ld, labels, n, k, counters)
}

}  // extern "C"
