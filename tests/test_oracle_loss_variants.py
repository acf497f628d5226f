"""CPU checks of oracle/loss_variants.py: every analytic gradient of the restated loss variants of
Training/train_base.py:156-574 and of minibatch discrimination against central differences in float64, and the algebraic relations
between the variants that the reference's expressions imply."""
import numpy as np
import pytest

from oracle import loss_variants as L
from oracle import tf_ops as T

N = dict(real=5, unl=7, gfake=6, bfake=4, d_real=6, d_fake=6)
VARIANTS = ['BGAN', 'GoodBadGAN', 'GoodRegBadGAN', 'GoodRegGAN', 'GoodRegGAN_cifar10', 'GoodRegGAN_BS', 'GoodRegGAN_BS_cifar10']
LAMBDA = [0.3, 0.7, 0.4, 0.25]


def make_inputs(seed=0, dtype=np.float64):
    r = np.random.default_rng(seed)
    g = lambda *s: r.standard_normal(s).astype(dtype)
    oh = lambda n: np.eye(10, dtype=dtype)[r.integers(0, 10, n)]
    return dict(d_real=g(N['d_real'], 1), d_fake=g(N['d_fake'], 1), d_unl=g(N['unl'], 1),
                c_real=g(N['real'], 10), c_unl=2 * g(N['unl'], 10), c_unl_d=g(3, 10), c_gfake=g(N['gfake'], 10), c_bfake=g(N['bfake'], 10),
                c_pert=g(N['bfake'], 10), c_rep=g(N['unl'], 10), c_unl_bg=g(N['unl'], 10), c_fake=g(N['bfake'], 10),
                f_real=g(N['real'], 16), f_unl=g(N['unl'], 16), f_bfake=g(N['bfake'], 16), f_pert=g(N['bfake'], 16), f_unl_bg=g(N['unl'], 16),
                y_l_c=oh(N['real']), y_g=oh(N['gfake']))


def call(variant, x, **kw):
    """-> (out, grads, {grads key: (input key, scalar picker)})"""
    D = [None, x['d_real'], None, x['d_fake'], None, x['d_unl']]
    Y = [x['y_g'], x['y_l_c']]
    first = lambda v: v[0] if isinstance(v, (list, tuple)) else v
    if variant == 'BGAN':
        out, g = L.loss_BGAN([x['c_real'], x['c_unl'], x['c_fake'], x['f_real'], x['f_unl'], x['f_bfake']], [x['y_l_c']])
        m = {'c_real': ('c_real', 1), 'c_unl': ('c_unl', 1), 'c_fake': ('c_fake', 1), 'feat_fake': ('f_bfake', 0)}
        return out, g, m
    if variant == 'GoodBadGAN':
        C = [x['c_real'], x['c_unl'], x['c_unl_d'], x['c_gfake'], x['c_bfake'], x['f_real'], x['f_unl'], x['f_bfake']]
        out, g = L.loss_GoodBadGAN(D, C, Y, LAMBDA[:1])
    elif variant == 'GoodRegBadGAN':
        C = [x['c_real'], x['c_unl'], x['c_unl_d'], x['c_gfake'], x['c_bfake'], x['c_pert'], x['f_real'], x['f_unl'], x['f_bfake'], x['f_pert']]
        out, g = L.loss_GoodRegBadGAN(D, C, Y, LAMBDA[:1])
    elif variant == 'GoodRegGAN':
        C = [x['c_real'], x['c_unl'], x['c_unl_d'], x['c_gfake'], x['c_bfake'], x['c_pert'], x['f_real'], x['f_unl'], x['f_bfake'], x['f_pert']]
        out, g = L.loss_GoodRegGAN(D, C, Y, LAMBDA[:3])
    elif variant == 'GoodRegGAN_cifar10':
        C = [x['c_real'], x['c_unl'], x['c_unl_d'], x['c_gfake'], x['c_bfake'], x['c_pert'], x['f_real'], x['f_unl'], x['f_bfake'], x['f_pert'],
             x['c_rep']]
        out, g = L.loss_GoodRegGAN_cifar10(D, C, Y, LAMBDA)
    elif variant == 'GoodRegGAN_BS':
        C = [x['c_real'], x['c_unl'], x['c_unl_d'], x['c_gfake'], x['c_bfake'], x['c_pert'], x['c_unl_bg'], x['f_real'], x['f_unl'], x['f_bfake'],
             x['f_pert'], x['f_unl_bg']]
        out, g = L.loss_GoodRegGAN_BS(D, C, Y, LAMBDA[:3])
    else:
        C = [x['c_real'], x['c_unl'], x['c_unl_d'], x['c_gfake'], x['c_bfake'], x['c_pert'], x['c_unl_bg'], x['f_real'], x['f_unl'], x['f_bfake'],
             x['f_pert'], x['f_unl_bg'], x['c_rep']]
        out, g = L.loss_GoodRegGAN_BS_cifar10(D, C, Y, LAMBDA, **kw)
    m = {'d_real': ('d_real', 0), 'd_fake': ('d_fake', 0), 'd_unl': ('d_unl', 0), 'gG_d_fake': ('d_fake', 1), 'feat_bfake': ('f_bfake', 2)}
    m.update({k: (k, 3) for k in g if k.startswith('c_')})
    return (first(out[0]), out[1], out[2], first(out[3])), g, m


@pytest.mark.parametrize("variant", VARIANTS)
def test_gradients_match_central_differences(variant):
    x = make_inputs(1)
    out, grads, m = call(variant, x)
    rng = np.random.default_rng(2)
    h = 1e-6
    for gk, (xk, which) in m.items():
        g = grads[gk]
        assert g.shape == x[xk].shape, (gk, g.shape, x[xk].shape)
        for _ in range(12):
            idx = tuple(rng.integers(0, s) for s in g.shape)
            xp, xm = dict(x), dict(x)
            xp[xk] = x[xk].copy(); xp[xk][idx] += h
            xm[xk] = x[xk].copy(); xm[xk][idx] -= h
            fd = (call(variant, xp)[0][which] - call(variant, xm)[0][which]) / (2 * h)
            assert abs(fd - g[idx]) <= 1e-6 * max(1.0, abs(fd)), (variant, gk, idx, fd, g[idx])


def test_relations_between_the_variants():
    x = make_inputs(3)
    # the perturbation term is the only difference between _loss_GoodBadGAN and _loss_GoodRegBadGAN (train_base.py:229 vs :566-569)
    a = call('GoodBadGAN', x)[0]
    b = call('GoodRegBadGAN', x)[0]
    assert a[:3] == b[:3]
    np.testing.assert_allclose(b[3] - a[3], 1e-3 * L.sqdiff_rows(x['c_pert'], x['c_bfake'])[0], rtol=1e-12)
    # _loss_GoodRegGAN_cifar10 = _loss_GoodRegGAN + lambda_4 * MSE (train_base.py:363-368)
    c = call('GoodRegGAN', x)[0]
    d = call('GoodRegGAN_cifar10', x)[0]
    np.testing.assert_allclose(d[3] - c[3], LAMBDA[3] * T.mse_mean(x['c_unl'], x['c_rep'])[0], rtol=1e-10)
    # FAST_MODE drops the generated-sample cross-entropy (train_base.py:475-478)
    e = call('GoodRegGAN_BS_cifar10', x)[0]
    f = call('GoodRegGAN_BS_cifar10', x, fast_mode=True)[0]
    np.testing.assert_allclose(e[3] - f[3], LAMBDA[1] * LAMBDA[0] * T.softmax_ce_mean(x['c_gfake'], x['y_g'])[0], rtol=1e-10)
    # every variant shares d_loss and the good generator's loss with _loss_GAN (train_base.py:123-128)
    assert a[0] == c[0] == e[0] and a[1] == c[1] == e[1]


def test_minibatch_discrimination_gradients():
    r = np.random.default_rng(5)
    n, f, k, d = 6, 9, 4, 5
    x, w, b = r.standard_normal((n, f)), 0.3 * r.standard_normal((f, k * d)), r.standard_normal(k)
    out, cache = L.minibatch_discrimination(x, w, b, d)
    assert out.shape == (n, k) and (out - b >= 1.0).all()               # the j = i term contributes exp(0)
    df = r.standard_normal((n, k))
    dx, dw, db = L.minibatch_discrimination_bwd(x, w, cache, df)
    loss = lambda x_, w_, b_: float((L.minibatch_discrimination(x_, w_, b_, d)[0] * df).sum())
    h = 1e-6
    for arr, g, pos in ((x, dx, 0), (w, dw, 1), (b, db, 2)):
        for _ in range(10):
            idx = tuple(r.integers(0, s) for s in arr.shape)
            args_p, args_m = [x, w, b], [x, w, b]
            ap, am = arr.copy(), arr.copy()
            ap[idx] += h
            am[idx] -= h
            args_p[pos], args_m[pos] = ap, am
            fd = (loss(*args_p) - loss(*args_m)) / (2 * h)
            assert abs(fd - g[idx]) <= 1e-6 * max(1.0, abs(fd)), (pos, idx, fd, g[idx])
