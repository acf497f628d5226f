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
                                                     float* __restrict__ loss, float* __restrict__ terms) {
  __shared__ float red[4];
  const int n = n_real + n_fake + n_unl;
  float acc = 0.f, tr[3] = {0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < n; r += 256) {
    const float v = z[(int64_t)r * ld];
    float t, w;
    int which;
    if (r < n_real) { t = 1.f; w = 1.f / n_real; which = 0; }
    else if (r < n_real + n_fake) { t = 0.f; w = 0.5f / n_fake; which = 1; }
    else { t = 0.f; w = 0.5f / n_unl; which = 2; }
    acc += w * bce(v, t);
    tr[which] += w * bce(v, t);
    float* o = dz + (int64_t)r * ld_d;
    o[0] = w * (sigm(v) - t);
    for (int k = 1; k < ld_d; ++k) o[k] = 0.f;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
  if (terms)                                             // {BCE(real,1), .5 BCE(fake,0), .5 BCE(unl,0)}
    for (int i = 0; i < 3; ++i) {
      const float sum = block_sum(tr[i], red);
      if (threadIdx.x == 0) terms[i] = sum;
    }
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

// term weights of the classifier loss: {CE(real), c_unl, H(unl), Bal(unl), CE(fake), MSE(unl,rep)}; the last two are used when the
// device pair `lam` is absent (_loss_GAN keeps lambda_1 / lambda_2 in device memory so that schedule changes reach replayed graphs)
struct CW { float real, unl, h, bal, fake, mse; };

// rows ordered [real | unl | unl_rep (n_rep = n_unl or 0) | fake (n_fake may be 0)]
// c_loss = w.unl*c_unl + w.real*CE(real) + w.h*H(unl) + w.bal*Bal(unl) + lam1*CE(fake) + lam2*MSE(unl,rep)
// (_loss_GAN: {1, 0.005, 1e-6, 1e-3}; the variants of train_base.py:156-574 differ in the weights only).  terms (optional): the six
// unweighted term values.
__global__ void __launch_bounds__(256) c_loss_kernel(const float* __restrict__ cl, int ld, int n_real, int n_unl, int n_rep, int n_fake,
                                                     const float* __restrict__ y_real, const float* __restrict__ y_fake, const float* __restrict__ d_unl,
                                                     int ld_dunl, const float* __restrict__ lam, CW w, float* __restrict__ dl, int ld_d,
                                                     float* __restrict__ loss, float* __restrict__ terms) {
  __shared__ float red[4];
  __shared__ float q[KC];
  const float lam1 = lam ? lam[0] : w.fake, lam2 = n_rep > 0 ? (lam ? lam[1] : w.mse) : 0.f;
  float t_real = 0.f, t_fake = 0.f, t_unl = 0.f, t_h = 0.f, t_mse = 0.f, t_bal = 0.f;
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
    acc += w.bal * bal;
    t_bal = bal;
  }
  // labelled rows (real, then fake)
  for (int pass = 0; pass < 2; ++pass) {
    const int n = pass == 0 ? n_real : n_fake, off = pass == 0 ? 0 : o_fake;
    const float* y = pass == 0 ? y_real : y_fake;
    const float wt = pass == 0 ? w.real : lam1;
    for (int r = threadIdx.x; r < n; r += 256) {
#pragma unroll
      for (int k = 0; k < KC; ++k) l[k] = cl[(int64_t)(off + r) * ld + k];
      softmax10(l, p, &lse);
      float ysum = 0.f, yl = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) { const float t = y[r * KC + k]; ysum += t; yl += t * l[k]; }
      acc += wt * (lse * ysum - yl) / n;
      if (pass == 0) t_real += (lse * ysum - yl) / n; else t_fake += (lse * ysum - yl) / n;
      float* o = dl + (int64_t)(off + r) * ld_d;
#pragma unroll
      for (int k = 0; k < KC; ++k) o[k] = wt * (p[k] * ysum - y[r * KC + k]) / n;
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
    const float rr = d_unl ? bce(d_unl[(int64_t)r * ld_dunl], 1.f) : 0.f;
    acc += w.unl * pm * rr / n_unl + w.h * (lse - pl) / n_unl;
    t_unl += pm * rr / n_unl;
    t_h += (lse - pl) / n_unl;
    float* o = dl + (int64_t)(o_unl + r) * ld_d;
    float* orep = dl + (int64_t)(o_rep + r) * ld_d;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      float g = w.unl * rr * pm * ((k == j ? 1.f : 0.f) - p[k]) / n_unl;      // C fools D (train_base.py:133-137)
      g += w.h * (p[k] - p[k] * (1.f + l[k] - pl)) / n_unl;                    // entropy
      g += w.bal * p[k] * (dq[k] - pdq);                                       // balance entropy
      if (n_rep > 0) {
        const float d = cl[(int64_t)(o_rep + r) * ld + k] - l[k];
        const float gm = lam2 * 2.f * d / (n_unl * KC);
        acc += lam2 * d * d / (n_unl * KC);
        t_mse += d * d / (n_unl * KC);
        g -= gm;
        orep[k] = gm;
      }
      o[k] = g;
    }
    for (int k = KC; k < ld_d; ++k) { o[k] = 0.f; if (n_rep > 0) orep[k] = 0.f; }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = acc;
  if (terms) {                                           // block-uniform
    const float v[6] = {t_real, t_unl, t_h, t_bal, t_fake, t_mse};
    for (int i = 0; i < 6; ++i) {
      const float sum = block_sum(v[i], red);
      if (threadIdx.x == 0) terms[i] = sum;
    }
  }
}

__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// bad-GAN "true-fake" terms of the classifier (train_base.py:162-166, 226-229, 296-298): with lse = logsumexp_k logits,
//   T_unl  = mean_unl(-0.5 lse + 0.5 softplus(lse)),   T_fake = 0.5 mean_fake softplus(lse);
// loss = {w_unl T_unl + w_fake T_fake, T_unl, T_fake}; d/dlogits = coefficient * softmax, written or (acc_* != 0) added.
__global__ void __launch_bounds__(256) true_fake_kernel(const float* __restrict__ unl, int ld_u, int n_unl, const float* __restrict__ fake, int ld_f,
                                                        int n_fake, float w_unl, float w_fake, float* __restrict__ d_unl, int ld_du, int acc_unl,
                                                        float* __restrict__ d_fake, int ld_df, int acc_fake, float* __restrict__ loss) {
  __shared__ float red[4];
  float l[KC], p[KC], lse, t_u = 0.f, t_f = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const int n = pass == 0 ? n_unl : n_fake;
    const float* src = pass == 0 ? unl : fake;
    const int ld = pass == 0 ? ld_u : ld_f, ldo = pass == 0 ? ld_du : ld_df, accum = pass == 0 ? acc_unl : acc_fake;
    float* dst = pass == 0 ? d_unl : d_fake;
    for (int r = threadIdx.x; r < n; r += 256) {
#pragma unroll
      for (int k = 0; k < KC; ++k) l[k] = src[(int64_t)r * ld + k];
      softmax10(l, p, &lse);
      float coef;
      if (pass == 0) { t_u += (-0.5f * lse + 0.5f * softplus(lse)) / n; coef = w_unl * (-0.5f + 0.5f * sigm(lse)) / n; }
      else { t_f += 0.5f * softplus(lse) / n; coef = w_fake * 0.5f * sigm(lse) / n; }
      float* o = dst + (int64_t)r * ldo;
#pragma unroll
      for (int k = 0; k < KC; ++k) o[k] = (accum ? o[k] : 0.f) + coef * p[k];
      if (!accum) for (int k = KC; k < ldo; ++k) o[k] = 0.f;
    }
  }
  t_u = block_sum(t_u, red);
  t_f = block_sum(t_f, red);
  if (threadIdx.x == 0) { loss[0] = w_unl * t_u + w_fake * t_f; loss[1] = t_u; loss[2] = t_f; }
}

// T = mean_n sum_k (a - b)^2 (train_base.py:299: consistency of the classifier under a perturbation of the bad generator's sample);
// loss = {w T, T}; da = 2 w (a - b) / n, db = -da, written or added.
__global__ void __launch_bounds__(256) sqdiff_rows_kernel(const float* __restrict__ a, int ld_a, const float* __restrict__ b, int ld_b, int n, int k,
                                                          float w, float* __restrict__ da, int ld_da, int acc_a, float* __restrict__ db, int ld_db,
                                                          int acc_b, float* __restrict__ loss) {
  __shared__ float red[4];
  float t = 0.f;
  for (int i = threadIdx.x; i < n * k; i += 256) {
    const int r = i / k, c = i - r * k;
    const float d = a[(int64_t)r * ld_a + c] - b[(int64_t)r * ld_b + c];
    t += d * d / n;
    const float g = 2.f * w * d / n;
    if (da) { float* o = da + (int64_t)r * ld_da + c; *o = (acc_a ? *o : 0.f) + g; }
    if (db) { float* o = db + (int64_t)r * ld_db + c; *o = (acc_b ? *o : 0.f) - g; }
  }
  if (!acc_a && da) for (int i = threadIdx.x; i < n * (ld_da - k); i += 256) da[(int64_t)(i / (ld_da - k)) * ld_da + k + i % (ld_da - k)] = 0.f;
  if (!acc_b && db) for (int i = threadIdx.x; i < n * (ld_db - k); i += 256) db[(int64_t)(i / (ld_db - k)) * ld_db + k + i % (ld_db - k)] = 0.f;
  t = block_sum(t, red);
  if (threadIdx.x == 0) { loss[0] = w * t; loss[1] = t; }
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

// Stand-alone heads behind the reference's Train_base helper methods (train_base.py:43-57,75-84): each is value + d/dlogits, the
// gradient written or (acc != 0) added so that a loss assembled from several helper calls accumulates into one gradient buffer.
// mean_n softmax-CE(labels, logits)  (tf.nn.softmax_cross_entropy_with_logits_v2 + reduce_mean); loss = {w T, T}
__global__ void __launch_bounds__(256) softmax_ce_kernel(const float* __restrict__ z, int ld, const float* __restrict__ y, int n, float w,
                                                         float* __restrict__ dz, int ld_d, int acc, float* __restrict__ loss) {
  __shared__ float red[4];
  float l[KC], p[KC], lse, t = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
#pragma unroll
    for (int k = 0; k < KC; ++k) l[k] = z[(int64_t)r * ld + k];
    softmax10(l, p, &lse);
    float ysum = 0.f, yl = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k) { const float tt = y[r * KC + k]; ysum += tt; yl += tt * l[k]; }
    t += (lse * ysum - yl) / n;
    if (dz) {
      float* o = dz + (int64_t)r * ld_d;
#pragma unroll
      for (int k = 0; k < KC; ++k) o[k] = (acc ? o[k] : 0.f) + w * (p[k] * ysum - y[r * KC + k]) / n;
      if (!acc) for (int k = KC; k < ld_d; ++k) o[k] = 0.f;
    }
  }
  t = block_sum(t, red);
  if (threadIdx.x == 0) { loss[0] = w * t; loss[1] = t; }
}

// mean over n*c elements of sigmoid-CE(labels, logits)  (tf.nn.sigmoid_cross_entropy_with_logits + reduce_mean); labels NULL: the
// constant `label` everywhere (tf.ones_like / tf.zeros_like, train_base.py:123-128); loss = {w T, T}
__global__ void __launch_bounds__(256) bce_logits_kernel(const float* __restrict__ z, int ld, const float* __restrict__ y, int ld_y, float label, int n,
                                                         int c, float w, float* __restrict__ dz, int ld_d, int acc, float* __restrict__ loss) {
  __shared__ float red[4];
  float t = 0.f;
  const float inv = 1.f / ((float)n * c);
  for (int i = threadIdx.x; i < n * c; i += 256) {
    const int r = i / c, k = i - r * c;
    const float v = z[(int64_t)r * ld + k], tt = y ? y[(int64_t)r * ld_y + k] : label;
    t += bce(v, tt) * inv;
    if (dz) { float* o = dz + (int64_t)r * ld_d + k; *o = (acc ? *o : 0.f) + w * (sigm(v) - tt) * inv; }
  }
  if (dz && !acc && ld_d > c)
    for (int i = threadIdx.x; i < n * (ld_d - c); i += 256) dz[(int64_t)(i / (ld_d - c)) * ld_d + c + i % (ld_d - c)] = 0.f;
  t = block_sum(t, red);
  if (threadIdx.x == 0) { loss[0] = w * t; loss[1] = t; }
}

// H = mean_n(lse - sum_k p_k l_k) (Train_base._entropy, train_base.py:43-48) and Bal = -sum_k (1/K) log(mean_n p_k + 1e-12)
// (_balance_entropy, :50-57); loss = {w_h H + w_bal Bal, H, Bal}
__global__ void __launch_bounds__(256) entropy_terms_kernel(const float* __restrict__ z, int ld, int n, float w_h, float w_bal, float* __restrict__ dz,
                                                            int ld_d, int acc, float* __restrict__ loss) {
  __shared__ float red[4];
  __shared__ float q[KC];
  float l[KC], p[KC], lse, qa[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) qa[k] = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
#pragma unroll
    for (int k = 0; k < KC; ++k) l[k] = z[(int64_t)r * ld + k];
    softmax10(l, p, &lse);
#pragma unroll
    for (int k = 0; k < KC; ++k) qa[k] += p[k];
  }
  for (int k = 0; k < KC; ++k) {
    const float s = block_sum(qa[k], red);
    if (threadIdx.x == 0) q[k] = s / n;
  }
  __syncthreads();
  float t_h = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
#pragma unroll
    for (int k = 0; k < KC; ++k) l[k] = z[(int64_t)r * ld + k];
    softmax10(l, p, &lse);
    float pl = 0.f, pdq = 0.f, dq[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) { pl += p[k] * l[k]; dq[k] = -1.f / (KC * (q[k] + 1e-12f)) / n; pdq += p[k] * dq[k]; }
    t_h += (lse - pl) / n;
    if (dz) {
      float* o = dz + (int64_t)r * ld_d;
#pragma unroll
      for (int k = 0; k < KC; ++k)
        o[k] = (acc ? o[k] : 0.f) + w_h * (p[k] - p[k] * (1.f + l[k] - pl)) / n + w_bal * p[k] * (dq[k] - pdq);
      if (!acc) for (int k = KC; k < ld_d; ++k) o[k] = 0.f;
    }
  }
  t_h = block_sum(t_h, red);
  if (threadIdx.x == 0) {
    float bal = 0.f;
    for (int k = 0; k < KC; ++k) bal -= logf(q[k] + 1e-12f) / KC;
    loss[0] = w_h * t_h + w_bal * bal; loss[1] = t_h; loss[2] = bal;
  }
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
  LOSS_LAUNCH(d_loss_kernel, logits, ld, n_real, n_fake, n_unl, dlogits, ld_d, loss, (float*)nullptr)
}

int tg_d_loss_terms_f32(const float* logits, int ld, int n_real, int n_fake, int n_unl, float* dlogits, int ld_d, float* loss, float* terms,
                        void* stream) {
  TG_REQUIRE(logits && dlogits && loss && terms && n_real > 0 && n_fake > 0 && n_unl > 0 && ld >= 1 && ld_d >= 1, "d_loss_terms: bad args");
  LOSS_LAUNCH(d_loss_kernel, logits, ld, n_real, n_fake, n_unl, dlogits, ld_d, loss, terms)
}

int tg_g_loss_f32(const float* logits, int ld, int n, float* dlogits, int ld_d, float* loss, void* stream) {
  TG_REQUIRE(logits && dlogits && loss && n > 0 && ld >= 1 && ld_d >= 1, "g_loss: bad args");
  LOSS_LAUNCH(g_loss_kernel, logits, ld, n, dlogits, ld_d, loss)
}

int tg_c_loss_f32(const float* c_logits, int ld, int n_real, int n_unl, int n_rep, int n_fake, const float* y_real, const float* y_fake,
                  const float* d_unl_logits, int ld_dunl, const float* lambdas, float* dlogits, int ld_d, float* loss, void* stream) {
  TG_REQUIRE(c_logits && y_real && y_fake && d_unl_logits && lambdas && dlogits && loss, "c_loss: null buffer");
  TG_REQUIRE(n_real > 0 && n_unl > 0 && n_fake > 0 && (n_rep == 0 || n_rep == n_unl) && ld >= KC && ld_d >= KC, "c_loss: bad sizes");
  const CW w{1.f, 0.005f, 1e-6f, 1e-3f, 0.f, 0.f};
  LOSS_LAUNCH(c_loss_kernel, c_logits, ld, n_real, n_unl, n_rep, n_fake, y_real, y_fake, d_unl_logits, ld_dunl, lambdas, w, dlogits, ld_d, loss,
              (float*)nullptr)
}

int tg_c_loss_terms_f32(const float* c_logits, int ld, int n_real, int n_unl, int n_rep, int n_fake, const float* y_real, const float* y_fake,
                        const float* d_unl_logits, int ld_dunl, const float* weights, float* dlogits, int ld_d, float* loss, float* terms,
                        void* stream) {
  TG_REQUIRE(c_logits && y_real && weights && dlogits && loss, "c_loss_terms: null buffer");
  TG_REQUIRE(n_real > 0 && n_unl > 0 && n_fake >= 0 && (n_rep == 0 || n_rep == n_unl) && ld >= KC && ld_d >= KC, "c_loss_terms: bad sizes");
  TG_REQUIRE(n_fake == 0 || y_fake, "c_loss_terms: y_fake is NULL with %d generated rows", n_fake);
  TG_REQUIRE(d_unl_logits || weights[1] == 0.f, "c_loss_terms: d_unl_logits is NULL but the C-fools-D weight is %g", (double)weights[1]);
  const CW w{weights[0], weights[1], weights[2], weights[3], weights[4], weights[5]};
  LOSS_LAUNCH(c_loss_kernel, c_logits, ld, n_real, n_unl, n_rep, n_fake, y_real, y_fake, d_unl_logits, ld_dunl, (const float*)nullptr, w, dlogits,
              ld_d, loss, terms)
}

int tg_true_fake_loss_f32(const float* unl_logits, int ld_u, int n_unl, const float* fake_logits, int ld_f, int n_fake, float w_unl, float w_fake,
                          float* d_unl, int ld_du, int accumulate_unl, float* d_fake, int ld_df, int accumulate_fake, float* loss, void* stream) {
  TG_REQUIRE(unl_logits && fake_logits && d_unl && d_fake && loss, "true_fake_loss: null buffer");
  TG_REQUIRE(n_unl > 0 && n_fake > 0 && ld_u >= KC && ld_f >= KC && ld_du >= KC && ld_df >= KC, "true_fake_loss: bad sizes");
  LOSS_LAUNCH(true_fake_kernel, unl_logits, ld_u, n_unl, fake_logits, ld_f, n_fake, w_unl, w_fake, d_unl, ld_du, accumulate_unl, d_fake, ld_df,
              accumulate_fake, loss)
}

int tg_sqdiff_rows_loss_f32(const float* a, int ld_a, const float* b, int ld_b, int n, int k, float w, float* da, int ld_da, int accumulate_a,
                            float* db, int ld_db, int accumulate_b, float* loss, void* stream) {
  TG_REQUIRE(a && b && loss && n > 0 && k > 0 && k <= ld_a && k <= ld_b, "sqdiff_rows_loss: bad args");
  TG_REQUIRE((!da || k <= ld_da) && (!db || k <= ld_db), "sqdiff_rows_loss: gradient stride smaller than k=%d", k);
  LOSS_LAUNCH(sqdiff_rows_kernel, a, ld_a, b, ld_b, n, k, w, da, ld_da, accumulate_a, db, ld_db, accumulate_b, loss)
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

int tg_softmax_ce_f32(const float* logits, int ld, const float* labels, int n, int k, float w, float* dlogits, int ld_d, int accumulate, float* loss,
                      void* stream) {
  TG_REQUIRE(logits && labels && loss && n > 0 && k == KC && ld >= KC && (!dlogits || ld_d >= KC), "softmax_ce: bad args (k must be %d)", KC);
  LOSS_LAUNCH(softmax_ce_kernel, logits, ld, labels, n, w, dlogits, ld_d, accumulate, loss)
}

int tg_bce_logits_f32(const float* logits, int ld, const float* labels, int ld_y, float label, int n, int c, float w, float* dlogits, int ld_d,
                      int accumulate, float* loss, void* stream) {
  TG_REQUIRE(logits && loss && n > 0 && c > 0 && c <= ld && (!labels || c <= ld_y) && (!dlogits || c <= ld_d), "bce_logits: bad args");
  LOSS_LAUNCH(bce_logits_kernel, logits, ld, labels, ld_y, label, n, c, w, dlogits, ld_d, accumulate, loss)
}

int tg_entropy_terms_f32(const float* logits, int ld, int n, int k, float w_h, float w_bal, float* dlogits, int ld_d, int accumulate, float* loss,
                         void* stream) {
  TG_REQUIRE(logits && loss && n > 0 && k == KC && ld >= KC && (!dlogits || ld_d >= KC), "entropy_terms: bad args (k must be %d)", KC);
  LOSS_LAUNCH(entropy_terms_kernel, logits, ld, n, w_h, w_bal, dlogits, ld_d, accumulate, loss)
}

int tg_accuracy_count_f32(const float* logits, int ld, const float* labels, int n, int k, float* counters, void* stream) {
  TG_REQUIRE(logits && labels && counters && n > 0 && k > 0 && k <= ld, "accuracy_count: bad args");
  LOSS_LAUNCH(accuracy_kernel, logits, ld, labels, n, k, counters)
}

}  // extern "C"
