"""Pin the oracle's primitives two independent ways: torch-CPU
re-derivation (test-only use of torch math, autograd for the backward) and float64
finite differences."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf_ops as T

torch.set_num_threads(4)


def t(x):
    return torch.tensor(x, dtype=torch.float64, requires_grad=True)


def tconv(x, w, stride, padding):
    """TF conv2d semantics on torch: NHWC/HWIO in, explicit asymmetric SAME pad."""
    n, h, wd, c = x.shape
    kh, kw = w.shape[:2]
    if padding == 'SAME':
        _, pt, pb = T.same_pad(h, kh, stride)
        _, pl, pr = T.same_pad(wd, kw, stride)
    else:
        pt = pb = pl = pr = 0
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xn, w.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)


CASES = [(2, 8, 8, 5, 7, 3, 1, 'SAME'), (2, 8, 8, 5, 7, 3, 2, 'SAME'), (2, 9, 7, 4, 6, 3, 2, 'SAME'),
         (3, 8, 8, 4, 6, 3, 1, 'VALID'), (2, 6, 6, 3, 4, 5, 2, 'SAME'), (2, 4, 4, 6, 5, 1, 1, 'SAME')]


@pytest.mark.parametrize("n,h,w,ci,co,k,s,pad", CASES)
def test_conv2d_and_grads(n, h, w, ci, co, k, s, pad):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, h, w, ci))
    wt = rng.standard_normal((k, k, ci, co))
    tx, tw = t(x), t(wt)
    ty = tconv(tx, tw, s, pad)
    y = T.conv2d(x, wt, (s, s), pad)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-12, atol=1e-12)
    dy = rng.standard_normal(y.shape)
    ty.backward(torch.tensor(dy))
    np.testing.assert_allclose(T.conv2d_bwd_filter(x, dy, wt.shape, (s, s), pad), tw.grad.numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(T.conv2d_bwd_input(x.shape, wt, dy, (s, s), pad), tx.grad.numpy(), rtol=1e-11, atol=1e-11)


def test_same_pad_extra_goes_after():
    assert T.same_pad(32, 3, 2) == (16, 0, 1)
    assert T.same_pad(32, 3, 1) == (32, 1, 1)
    assert T.same_pad(8, 5, 2) == (4, 1, 2)


def test_conv2d_transpose_alignment_and_grads():
    """out[2i+k-1] += in[i]*W[k], filter [kh,kw,Cout,Cin]."""
    rng = np.random.default_rng(1)
    n, h, cin, cout = 2, 4, 6, 3
    x = rng.standard_normal((n, h, h, cin))
    w = rng.standard_normal((5, 5, cout, cin))
    y = T.conv2d_transpose(x, w)
    assert y.shape == (n, 2 * h, 2 * h, cout)
    ref = np.zeros_like(y)
    for iy in range(h):
        for ix in range(h):
            for ky in range(5):
                for kx in range(5):
                    oy, ox = 2 * iy + ky - 1, 2 * ix + kx - 1
                    if 0 <= oy < 2 * h and 0 <= ox < 2 * h:
                        ref[:, oy, ox, :] += x[:, iy, ix, :] @ w[ky, kx].T
    np.testing.assert_allclose(y, ref, rtol=1e-12, atol=1e-12)
    # independent: transpose == autograd of the forward strided conv
    X = torch.zeros((n, 2 * h, 2 * h, cout), dtype=torch.float64, requires_grad=True)
    tw, tx = t(w), t(x)
    fwd = tconv(X, tw, 2, 'SAME')
    (gy,) = torch.autograd.grad(fwd, X, tx, create_graph=True)
    np.testing.assert_allclose(y, gy.detach().numpy(), rtol=1e-12, atol=1e-12)
    dy = rng.standard_normal(y.shape)
    gy.backward(torch.tensor(dy))
    np.testing.assert_allclose(T.conv2d_transpose_bwd_filter(x, dy, w.shape), tw.grad.numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(T.conv2d_transpose_bwd_input(w, dy), tx.grad.numpy(), rtol=1e-11, atol=1e-11)


def test_wn_weight_bwd():
    rng = np.random.default_rng(2)
    for shape, ax in (((3, 3, 4, 5), -1), ((7, 5), -1), ((5, 5, 3, 6), 2)):
        v, g = rng.standard_normal(shape), rng.standard_normal(shape[ax])
        dw = rng.standard_normal(shape)
        tv, tg = t(v), t(g)
        axes = [a for a in range(len(shape)) if a != ax % len(shape)]
        shp = [1] * len(shape)
        shp[ax] = -1
        tw = tg.reshape(shp) * tv / torch.sqrt((tv * tv).sum(dim=axes, keepdim=True))
        np.testing.assert_allclose(T.wn_weight(v, g, ax), tw.detach().numpy(), rtol=1e-12)
        tw.backward(torch.tensor(dw))
        dv, dg = T.wn_weight_bwd(v, g, dw, ax)
        np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(dg, tg.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_norms_bwd():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((4, 5, 5, 6))
    gam, bet, dy = rng.standard_normal(6), rng.standard_normal(6), rng.standard_normal(x.shape)
    tx, tg, tb = t(x), t(gam), t(bet)
    mu = tx.mean(dim=(0, 1, 2))
    var = ((tx - mu) ** 2).mean(dim=(0, 1, 2))
    ty = tg * (tx - mu) / torch.sqrt(var + 1e-5) + tb
    y, cache = T.batch_norm_train(x, gam, bet, 1e-5)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-11, atol=1e-12)
    ty.backward(torch.tensor(dy))
    dx, dg, db = T.batch_norm_train_bwd(dy, gam, cache)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dg, tg.grad.numpy(), rtol=1e-10)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-10)
    # mean-only BN
    tx = t(x)
    ty = tx - tx.mean(dim=(0, 1, 2)) + tb.detach()
    y2, pop = T.mobn_train(x, np.zeros(6), bet)
    np.testing.assert_allclose(y2, ty.detach().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pop, 0.1 * x.mean(axis=(0, 1, 2)), rtol=1e-12)
    ty.backward(torch.tensor(dy))
    np.testing.assert_allclose(T.mobn_train_bwd(dy)[0], tx.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_pools():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3, 8, 8, 5))
    tx = t(x)
    ty = F.max_pool2d(tx.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    y, idx = T.maxpool2(x)
    np.testing.assert_array_equal(y, ty.detach().numpy())
    dy = rng.standard_normal(y.shape)
    ty.backward(torch.tensor(dy))
    np.testing.assert_array_equal(T.maxpool2_bwd(dy, idx, x.shape), tx.grad.numpy())
    tx = t(x)
    tg = tx.amax(dim=(1, 2))
    g, gi = T.global_maxpool(x)
    np.testing.assert_array_equal(g, tg.detach().numpy())
    dg = rng.standard_normal(g.shape)
    tg.backward(torch.tensor(dg))
    np.testing.assert_array_equal(T.global_maxpool_bwd(dg, gi, x.shape), tx.grad.numpy())


def _tsm(z):
    return torch.softmax(z, dim=1)


def test_losses_match_torch_autograd():
    rng = np.random.default_rng(5)
    n = 12
    lc, lr_ = rng.standard_normal((n, 10)) * 3, rng.standard_normal((n, 10)) * 3
    ld = rng.standard_normal((n, 1)) * 2
    lab = np.eye(10)[rng.integers(0, 10, n)]

    def check(val, grad, tval, tz):
        tval.backward()
        np.testing.assert_allclose(val, tval.item(), rtol=1e-11)
        np.testing.assert_allclose(grad, tz.grad.numpy(), rtol=1e-9, atol=1e-13)

    z = t(ld)
    check(*T.bce_mean(ld, np.ones_like(ld)), F.binary_cross_entropy_with_logits(z, torch.ones_like(z)), z)
    z = t(ld)
    check(*T.bce_mean(ld, np.zeros_like(ld)), F.binary_cross_entropy_with_logits(z, torch.zeros_like(z)), z)
    z = t(lc)
    check(*T.softmax_ce_mean(lc, lab), -(torch.tensor(lab) * torch.log_softmax(z, 1)).sum(1).mean(), z)
    z = t(lc)
    p = _tsm(z)
    check(*T.entropy(lc), (-(p * z).sum(1) + torch.logsumexp(z, 1)).mean(), z)
    z = t(lc)
    check(*T.balance_entropy(lc), -(torch.log(_tsm(z).mean(0) + 1e-12) / 10).sum(), z)
    z = t(lc)
    r = F.binary_cross_entropy_with_logits(torch.tensor(ld), torch.ones((n, 1), dtype=torch.float64), reduction='none').mean(1)
    check(*T.c_unl_loss(lc, ld), (_tsm(z).max(dim=1).values * r).mean(), z)
    za, zb = t(lc), t(lr_)
    tv = ((za - zb) ** 2).mean()
    tv.backward()
    v, ga, gb = T.mse_mean(lc, lr_)
    np.testing.assert_allclose(v, tv.item(), rtol=1e-12)
    np.testing.assert_allclose(ga, za.grad.numpy(), rtol=1e-10)
    np.testing.assert_allclose(gb, zb.grad.numpy(), rtol=1e-10)


def test_fm_and_pull_away():
    rng = np.random.default_rng(6)
    ff, fu = rng.standard_normal((9, 16)), rng.standard_normal((7, 16))
    a, b = t(ff), t(fu)
    tv = (a.mean(0) - b.mean(0)).abs().mean()
    tv.backward()
    v, ga, gb = T.feature_match(ff, fu)
    np.testing.assert_allclose(v, tv.item(), rtol=1e-12)
    np.testing.assert_allclose(ga, a.grad.numpy(), rtol=1e-10)
    np.testing.assert_allclose(gb, b.grad.numpy(), rtol=1e-10)
    a = t(ff)
    fn = a / a.norm(dim=1, keepdim=True)
    c = fn @ fn.T
    mask = 1 - torch.eye(9, dtype=torch.float64)
    tv = 0.8 * ((c * mask) ** 2).sum() / (9 * 8)
    tv.backward()
    v, g = T.pull_away_masked(ff)
    np.testing.assert_allclose(v, tv.item(), rtol=1e-12)
    np.testing.assert_allclose(g, a.grad.numpy(), rtol=1e-9, atol=1e-13)
    a = t(ff)
    fn = a / a.norm(dim=1, keepdim=True)
    tv = 0.8 * (fn @ fn.T).mean()
    tv.backward()
    v, g = T.pull_away_unmasked(ff)
    np.testing.assert_allclose(v, tv.item(), rtol=1e-12)
    np.testing.assert_allclose(g, a.grad.numpy(), rtol=1e-9, atol=1e-13)


def test_adam_is_tf_form():
    rng = np.random.default_rng(7)
    p, g = rng.standard_normal(50), rng.standard_normal(50)
    m = v = np.zeros(50)
    ref = p.copy()
    for step in range(1, 4):
        p, m, v = T.adam_update(p, g, m, v, step, 3e-4, 0.5)
    tp = torch.tensor(ref, requires_grad=True)
    # torch Adam differs only in where eps enters; with eps_hat = eps*sqrt(1-b2^t) they coincide,
    # so compare against the closed form instead.
    mm = vv = np.zeros(50)
    q = ref.copy()
    for step in range(1, 4):
        mm = 0.5 * mm + 0.5 * g
        vv = 0.999 * vv + 0.001 * g * g
        q = q - 3e-4 * np.sqrt(1 - 0.999 ** step) / (1 - 0.5 ** step) * mm / (np.sqrt(vv) + 1e-8)
    np.testing.assert_allclose(p, q, rtol=1e-12)


def test_dropout_and_concat():
    x = np.arange(24, dtype=np.float64).reshape(1, 2, 3, 4)
    mask = (np.arange(24).reshape(1, 2, 3, 4) % 2).astype(np.float64)
    np.testing.assert_allclose(T.dropout(x, mask, 0.2), x * mask / 0.8)
    y = np.eye(10)[[3]]
    cc = T.conv_cond_concat(x, y)
    assert cc.shape == (1, 2, 3, 14)
    assert (cc[0, :, :, 4 + 3] == 1).all() and cc[0, :, :, 4:].sum() == 6
