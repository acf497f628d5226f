"""ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

CPU (NumPy) restatement of the TensorFlow-1.x ops the reference's Triple-GAN hot
path executes, forward AND hand-derived backward.  The product path
(`tensorflow-implementation-of-triple-gan_amd/`) never imports this module.

PARITY UNPINNED: the reference holds no tests / golden vectors and TensorFlow 1.x
is not importable in the build container, so these functions are
pinned only by (i) an independent torch-CPU re-derivation in tests/test_oracle_*.py
and (ii) float64 finite-difference gradient checks.

Every function cites the reference call site (paths relative to /root/reference)
whose TF op it restates.  All functions are dtype-agnostic (float32 for parity,
float64 for gradient checks).  Layouts follow the reference: NHWC activations,
HWIO conv filters, [kh,kw,Cout,Cin] transposed-conv filters, [in,out] dense.
"""
import numpy as np

# --------------------------------------------------------------------------
# padding arithmetic (TF "SAME": extra pixel goes AFTER) — SURVEY App. C.1
# --------------------------------------------------------------------------

def same_pad(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def _out_and_pad(h, w, kh, kw, sh, sw, padding):
    if padding == 'SAME':
        ho, pt, pb = same_pad(h, kh, sh)
        wo, pl, pr = same_pad(w, kw, sw)
    elif padding == 'VALID':
        ho, wo = (h - kh) // sh + 1, (w - kw) // sw + 1
        pt = pb = pl = pr = 0
    else:
        raise ValueError(padding)
    return ho, wo, pt, pb, pl, pr


def _patches(xp, kh, kw, sh, sw, ho, wo):
    """[N,Hp,Wp,C] -> [N*ho*wo, kh*kw*C] (copy)."""
    n, _, _, c = xp.shape
    s0, s1, s2, s3 = xp.strides
    v = np.lib.stride_tricks.as_strided(
        xp, shape=(n, ho, wo, kh, kw, c),
        strides=(s0, s1 * sh, s2 * sw, s1, s2, s3), writeable=False)
    return v.reshape(n * ho * wo, kh * kw * c)


_CHUNK_ELEMS = 48 * 1024 * 1024  # bound the im2col scratch (floats)


# --------------------------------------------------------------------------
# "bf16 MFMA conv path" (BASELINE.json configs[3]) — build-defined, not in the reference: every operand of a
# conv / transposed-conv / dense contraction (activation, effective filter, output gradient) is rounded to bfloat16
# (float32 value, round-to-nearest-even on bit 16) and the products are accumulated in the array dtype.  Enabled
# for a whole step by MFMA_BF16 = True (tests only), or per call through bf16_round().
# --------------------------------------------------------------------------
MFMA_BF16 = False


def bf16_round(x):
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    u = x32.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(np.asarray(x).dtype)


def _q(x):
    return bf16_round(x) if MFMA_BF16 else x


# --------------------------------------------------------------------------
# Summation-order control (tests/golden/make_golden_long.py): with SUM_REVERSED = True every contraction of a conv / transposed
# conv / dense product and of their two gradients runs over its reduction index BACKWARDS (operands are copied reversed before the
# BLAS call).  Mathematically the same numbers; in float32 a second, equally accurate rounding of each of them.  The long-horizon
# fixtures use it to measure how far two correct float32 evaluations of the SAME free-running trajectory drift apart.
# --------------------------------------------------------------------------
SUM_REVERSED = False


def _mm(a, b):
    """a @ b over the reduction index in the order SUM_REVERSED selects."""
    if SUM_REVERSED:
        return np.ascontiguousarray(a[:, ::-1]) @ np.ascontiguousarray(b[::-1])
    return a @ b


def matmul(a, b):
    """a @ b with the operand rounding of the bf16 MFMA path when it is switched on (dense layers of the nets)."""
    return _mm(_q(a), _q(b))


def _chunks(n, per_image_elems):
    step = max(1, _CHUNK_ELEMS // max(1, per_image_elems))
    for i in range(0, n, step):
        yield i, min(n, i + step)


# --------------------------------------------------------------------------
# tf.nn.conv2d and its two gradients — Model/nn.py:504, Model/modle_base.py:102,161
# --------------------------------------------------------------------------

def conv2d(x, w, stride=(1, 1), padding='SAME'):
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    assert ci == c
    sh, sw = stride
    ho, wo, pt, pb, pl, pr = _out_and_pad(h, wd, kh, kw, sh, sw, padding)
    xp = np.pad(_q(x), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    w2 = _q(w).reshape(kh * kw * ci, co)
    y = np.empty((n, ho, wo, co), x.dtype)
    for a, b in _chunks(n, ho * wo * kh * kw * c):
        y[a:b] = _mm(_patches(xp[a:b], kh, kw, sh, sw, ho, wo), w2).reshape(b - a, ho, wo, co)
    return y


def conv2d_bwd_filter(x, dy, wshape, stride=(1, 1), padding='SAME'):
    n, h, wd, c = x.shape
    kh, kw, ci, co = wshape
    sh, sw = stride
    ho, wo, pt, pb, pl, pr = _out_and_pad(h, wd, kh, kw, sh, sw, padding)
    assert dy.shape == (n, ho, wo, co)
    xp = np.pad(_q(x), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dy = _q(dy)
    dw = np.zeros((kh * kw * ci, co), x.dtype)
    for a, b in _chunks(n, ho * wo * kh * kw * c):
        dw += _mm(_patches(xp[a:b], kh, kw, sh, sw, ho, wo).T, dy[a:b].reshape(-1, co))
    return dw.reshape(kh, kw, ci, co)


def conv2d_bwd_input(xshape, w, dy, stride=(1, 1), padding='SAME'):
    n, h, wd, c = xshape
    kh, kw, ci, co = w.shape
    sh, sw = stride
    ho, wo, pt, pb, pl, pr = _out_and_pad(h, wd, kh, kw, sh, sw, padding)
    assert dy.shape == (n, ho, wo, co), (dy.shape, (n, ho, wo, co))
    dxp = np.zeros((n, h + pt + pb, wd + pl + pr, c), dy.dtype)
    dy = _q(dy)
    w2t = _q(w).reshape(kh * kw * ci, co).T
    for a, b in _chunks(n, ho * wo * kh * kw * c):
        dp = _mm(dy[a:b].reshape(-1, co), w2t).reshape(b - a, ho, wo, kh, kw, ci)
        for ky in range(kh):
            for kx in range(kw):
                dxp[a:b, ky:ky + sh * ho:sh, kx:kx + sw * wo:sw, :] += dp[:, :, :, ky, kx, :]
    return dxp[:, pt:pt + h, pl:pl + wd, :]


# --------------------------------------------------------------------------
# tf.layers.conv2d_transpose / tf.nn.conv2d_transpose 'same' — modle_base.py:149,250
# filter [kh,kw,Cout,Cin]; out = 2x in for stride 2.  It IS the input-gradient of
# the forward conv (Cout -> Cin) with the same filter read as HWIO. App. C.2.
# --------------------------------------------------------------------------

def conv2d_transpose(x, w, stride=(2, 2), padding='SAME'):
    n, h, wd, cin = x.shape
    kh, kw, cout, ci = w.shape
    assert ci == cin
    sh, sw = stride
    assert padding == 'SAME'
    return conv2d_bwd_input((n, h * sh, wd * sw, cout), w, x, stride, padding)


def conv2d_transpose_bwd_input(w, dy, stride=(2, 2), padding='SAME'):
    return conv2d(dy, w, stride, padding)


def conv2d_transpose_bwd_filter(x, dy, wshape, stride=(2, 2), padding='SAME'):
    return conv2d_bwd_filter(dy, x, wshape, stride, padding)


# --------------------------------------------------------------------------
# weight normalisation — nn.py:502 (conv, axes 0,1,2), nn.py:554 (dense, NO eps),
# modle_base.py:66,101,148
# --------------------------------------------------------------------------

def l2_normalize(v, axes, eps=1e-12):
    """tf.nn.l2_normalize: v * rsqrt(max(sum v^2, eps))."""
    ss = np.sum(np.square(v), axis=tuple(axes), keepdims=True)
    return v / np.sqrt(np.maximum(ss, np.asarray(eps, v.dtype)))


def wn_weight(v, g, out_axis=-1):
    """W = g * V/||V|| with the norm over every axis except `out_axis`."""
    out_axis %= v.ndim
    axes = tuple(a for a in range(v.ndim) if a != out_axis)
    shp = [1] * v.ndim
    shp[out_axis] = -1
    return g.reshape(shp) * l2_normalize(v, axes)


def wn_weight_bwd(v, g, dw, out_axis=-1):
    """Gradients of W = g V/||V|| (eps branch of l2_normalize never active)."""
    out_axis %= v.ndim
    axes = tuple(a for a in range(v.ndim) if a != out_axis)
    shp = [1] * v.ndim
    shp[out_axis] = -1
    nrm = np.sqrt(np.sum(np.square(v), axis=axes, keepdims=True))
    vhat = v / nrm
    dot = np.sum(dw * vhat, axis=axes, keepdims=True)
    dg = dot.reshape(-1)
    dv = (g.reshape(shp) / nrm) * (dw - vhat * dot)
    return dv, dg


# --------------------------------------------------------------------------
# mean-only batch norm — nn.py:147-187
# --------------------------------------------------------------------------

def mobn_train(x, pop_mean, b, decay=0.9):
    axes = tuple(range(x.ndim - 1))
    m = x.mean(axis=axes, dtype=x.dtype)
    new_pop = pop_mean * decay + m * (1 - decay)
    return x - m + b, new_pop.astype(x.dtype)


def mobn_eval(x, pop_mean, b):
    return x - pop_mean + b


def mobn_train_bwd(dy):
    axes = tuple(range(dy.ndim - 1))
    db = dy.sum(axis=axes)
    dx = dy - dy.mean(axis=axes)
    return dx, db


# --------------------------------------------------------------------------
# tf.contrib.layers.batch_norm(decay, eps, scale=True, is_training=True,
# updates_collections=None) — modle_base.py:229-237
# --------------------------------------------------------------------------

def batch_norm_train(x, gamma, beta, eps=1e-5):
    axes = tuple(range(x.ndim - 1))
    mu = x.mean(axis=axes)
    var = np.mean(np.square(x - mu), axis=axes)          # biased
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * inv
    return gamma * xhat + beta, (xhat, inv, mu, var)


def batch_norm_train_bwd(dy, gamma, cache):
    xhat, inv, _, _ = cache
    axes = tuple(range(dy.ndim - 1))
    dgamma = np.sum(dy * xhat, axis=axes)
    dbeta = np.sum(dy, axis=axes)
    m = dy.size // dy.shape[-1]
    dx = (gamma * inv) * (dy - dbeta / m - xhat * (dgamma / m))
    return dx, dgamma, dbeta


def batch_norm_moving_update(mm, mv, mu, var, count, decay=0.9, fused=True):
    """moving stats (dead state: G's BN always runs in training mode, App. C.5).
    The fused (4-D) kernel feeds the Bessel-corrected variance [UNVERIFIED-TF]."""
    v = var * (count / max(count - 1, 1)) if fused else var
    return mm * decay + mu * (1 - decay), mv * decay + v * (1 - decay)


# --------------------------------------------------------------------------
# pointwise — Good_GAN_cifar10.py:26-27, modle_base.py:178-185
# --------------------------------------------------------------------------

def relu(x):
    return np.maximum(x, 0)


def lrelu(x, alpha=0.2):
    """relu(x) - alpha*relu(-x)."""
    return np.where(x > 0, x, alpha * x).astype(x.dtype)


def lrelu_bwd_from_out(y, dy, alpha=0.2):
    return np.where(y > 0, dy, alpha * dy).astype(dy.dtype)


def relu_bwd_from_out(y, dy):
    return np.where(y > 0, dy, 0).astype(dy.dtype)


def softplus(x):
    return np.logaddexp(x, 0).astype(x.dtype)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


# --------------------------------------------------------------------------
# dropout / noise / concat — modle_base.py:190-202,239-244.  App. C.4
# --------------------------------------------------------------------------

def dropout(x, mask, rate):
    """mask is the 0/1 keep mask (floor(keep + U[0,1)))."""
    keep = 1.0 - rate
    return x * mask * np.asarray(1.0 / keep, x.dtype)


def dropout_bwd(dy, mask, rate):
    return dropout(dy, mask, rate)


def conv_cond_concat(x, y):
    n, h, w, _ = x.shape
    yb = np.broadcast_to(y.reshape(n, 1, 1, -1), (n, h, w, y.shape[-1]))
    return np.concatenate([x, yb.astype(x.dtype)], axis=3)


# --------------------------------------------------------------------------
# pooling — Good_GAN_cifar10.py:94,123,142,163
# --------------------------------------------------------------------------

def maxpool2(x):
    n, h, w, c = x.shape
    v = x.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, c)
    idx = v.argmax(axis=3)                                # first max wins
    return v.max(axis=3), idx


def maxpool2_bwd(dy, idx, xshape):
    n, h, w, c = xshape
    d = np.zeros((n, h // 2, w // 2, 4, c), dy.dtype)
    np.put_along_axis(d, idx[:, :, :, None, :], dy[:, :, :, None, :], axis=3)
    return d.reshape(n, h // 2, w // 2, 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, c)


def global_maxpool(x):
    n, h, w, c = x.shape
    v = x.reshape(n, h * w, c)
    idx = v.argmax(axis=1)
    return v.max(axis=1), idx


def global_maxpool_bwd(dy, idx, xshape):
    n, h, w, c = xshape
    d = np.zeros((n, h * w, c), dy.dtype)
    np.put_along_axis(d, idx[:, None, :], dy[:, None, :], axis=1)
    return d.reshape(n, h, w, c)


def global_avgpool(x):
    return x.mean(axis=(1, 2))


def global_avgpool_bwd(dy, xshape):
    n, h, w, c = xshape
    return np.broadcast_to(dy[:, None, None, :] / (h * w), xshape).astype(dy.dtype)


def argmax_onehot(logits, depth=10):
    return np.eye(depth, dtype=logits.dtype)[np.argmax(logits, axis=1)]


# --------------------------------------------------------------------------
# losses — Training/train_base.py:43-57,75-84,113-154 (+172-182,202-207)
# each returns (value, gradient wrt logits)
# --------------------------------------------------------------------------

def _softmax(z):
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def _logsumexp(z):
    m = z.max(axis=1, keepdims=True)
    return (m + np.log(np.exp(z - m).sum(axis=1, keepdims=True)))[:, 0]


def bce_rows(z, t):
    """tf.nn.sigmoid_cross_entropy_with_logits elementwise."""
    return np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))


def bce_mean(z, t):
    """train_base.py:81-84; grad wrt z."""
    return bce_rows(z, t).mean(), (sigmoid(z) - t) / z.size


def softmax_ce_mean(logits, labels):
    """train_base.py:75-79."""
    p = _softmax(logits)
    loss = (_logsumexp(logits) * labels.sum(axis=1) - (labels * logits).sum(axis=1)).mean()
    return loss, (p * labels.sum(axis=1, keepdims=True) - labels) / logits.shape[0]


def entropy(logits):
    """train_base.py:43-48: mean_n(logsumexp - sum_k p_k l_k)."""
    p = _softmax(logits)
    pl = (p * logits).sum(axis=1, keepdims=True)
    val = (_logsumexp(logits) - pl[:, 0]).mean()
    grad = (p - p * (1 + logits - pl)) / logits.shape[0]
    return val, grad


def balance_entropy(logits):
    """train_base.py:50-57: -sum_k (1/K) log(mean_n p_k + 1e-12)."""
    n, k = logits.shape
    p = _softmax(logits)
    q = p.mean(axis=0)
    val = -np.sum(np.log(q + 1e-12) / k)
    dq = -1.0 / (k * (q + 1e-12)) / n                      # d val / d p[n,k]
    grad = p * (dq - (p * dq).sum(axis=1, keepdims=True))
    return val, grad


def c_unl_loss(c_unl_logits, d_unl_logits):
    """train_base.py:133-137: mean_n( max softmax(C_unl)_n * BCE(D_unl_n, 1) ); grad wrt C_unl."""
    n = c_unl_logits.shape[0]
    p = _softmax(c_unl_logits)
    j = p.argmax(axis=1)
    pm = p[np.arange(n), j]
    r = bce_rows(d_unl_logits, np.ones_like(d_unl_logits)).mean(axis=1)
    val = (pm * r).mean()
    oh = np.zeros_like(p)
    oh[np.arange(n), j] = 1
    grad = (r * pm)[:, None] * (oh - p) / n
    return val, grad


def mse_mean(a, b):
    """tf.losses.mean_squared_error(labels=a, predictions=b); grads wrt BOTH (App. C.11)."""
    d = b - a
    val = np.mean(np.square(d))
    gb = 2 * d / d.size
    return val, -gb, gb


def feature_match(f_fake, f_unl):
    """train_base.py:172: mean_c |mean_n f_fake - mean_n f_unl|; grads wrt both."""
    d = f_fake.mean(axis=0) - f_unl.mean(axis=0)
    val = np.abs(d).mean()
    s = np.sign(d) / d.size
    return val, np.broadcast_to(s / f_fake.shape[0], f_fake.shape).copy(), \
        np.broadcast_to(-s / f_unl.shape[0], f_unl.shape).copy()


def pull_away_masked(f):
    """train_base.py:175-181: 0.8 * sum_{i!=j} cos^2(f_i,f_j) / (N(N-1)); grad wrt f."""
    n = f.shape[0]
    nr = np.sqrt(np.sum(f * f, axis=1, keepdims=True))
    fn = f / nr
    c = fn @ fn.T
    mask = 1.0 - np.eye(n, dtype=f.dtype)
    val = 0.8 * np.sum(np.square(c * mask)) / (n * (n - 1))
    dc = 0.8 * 2 * c * mask / (n * (n - 1))
    dfn = (dc + dc.T) @ fn
    df = (dfn - fn * np.sum(dfn * fn, axis=1, keepdims=True)) / nr
    return val, df


def pull_away_unmasked(f):
    """train_base.py:204-207: 0.8 * mean_ij cos(f_i,f_j); grad wrt f."""
    n = f.shape[0]
    nr = np.sqrt(np.sum(f * f, axis=1, keepdims=True))
    fn = f / nr
    val = 0.8 * np.mean(fn @ fn.T)
    dfn = 0.8 * 2 * fn.sum(axis=0, keepdims=True) / (n * n) * np.ones_like(fn)
    df = (dfn - fn * np.sum(dfn * fn, axis=1, keepdims=True)) / nr
    return val, df


# --------------------------------------------------------------------------
# optimiser — train_base.py:91-97 (tf.train.AdamOptimizer), Train_goodGAN.py:101-103 (EMA)
# --------------------------------------------------------------------------

def adam_update(p, g, m, v, t, lr, beta1, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer as its ApplyAdam functor computes it, in the variable's dtype [UNVERIFIED-TF]:
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); p -= m*alpha/(sqrt(v)+eps).
    t counts from 1.  (1-b) is formed in the variable's precision, as T(1)-beta does.)"""
    f = p.dtype.type
    alpha = f(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    m = m + (g - m) * (f(1) - f(beta1))
    v = v + (g * g - v) * (f(1) - f(beta2))
    p = p - m * alpha / (np.sqrt(v) + f(eps))
    return p, m, v


def ema_update(shadow, p, decay=0.9999):
    f = p.dtype.type
    return shadow - f(1 - decay) * (shadow - p)
