"""GPU parity of the standalone kernels behind the C ABI: loss heads (+ feature-matching / pull-away), Adam / EMA,
argmax one-hot, accuracy counter, Philox RNG statistics, segmented column statistics."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _lib():
    from tg import lib
    lib.load()
    return lib


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_d_and_g_loss():
    lib = _lib()
    rng = np.random.default_rng(0)
    n_r, n_f, n_u = 100, 100, 50
    z = (rng.standard_normal((n_r + n_f + n_u, 1)) * 3).astype(np.float32)
    lr, gr = T.bce_mean(z[:n_r], np.ones((n_r, 1), np.float32))
    lf, gf = T.bce_mean(z[n_r:n_r + n_f], np.zeros((n_f, 1), np.float32))
    lu, gu = T.bce_mean(z[n_r + n_f:], np.zeros((n_u, 1), np.float32))
    zd, dz, loss = dev(z), torch.full((250, 32), 7.0, device='cuda'), torch.zeros(1, device='cuda')
    lib.call('tg_d_loss_f32', lib.ptr(zd), 1, n_r, n_f, n_u, lib.ptr(dz), 32, lib.ptr(loss), st())
    assert abs(loss.item() - (lr + 0.5 * lf + 0.5 * lu)) < 1e-5
    g = dz.cpu().numpy()
    np.testing.assert_allclose(g[:, 0:1], np.concatenate([gr, 0.5 * gf, 0.5 * gu]), rtol=1e-4, atol=1e-8)
    assert (g[:, 1:] == 0).all()
    lib.call('tg_g_loss_f32', lib.ptr(zd), 1, n_r, lib.ptr(dz), 32, lib.ptr(loss), st())
    l1, g1 = T.bce_mean(z[:n_r], np.ones((n_r, 1), np.float32))
    assert abs(loss.item() - 0.5 * l1) < 1e-5
    np.testing.assert_allclose(dz.cpu().numpy()[:n_r, 0:1], 0.5 * g1, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("with_rep", [True, False])
def test_c_loss(with_rep):
    lib = _lib()
    rng = np.random.default_rng(1)
    n_r, n_u, n_f = 50, 50, 100
    lam1, lam2 = 0.3, 0.5
    c_real, c_unl, c_rep, c_fake = [(rng.standard_normal((n, 10)) * 2).astype(np.float32) for n in (n_r, n_u, n_u, n_f)]
    y_r = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n_r)]
    y_f = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n_f)]
    d_unl = (rng.standard_normal((n_u, 1)) * 2).astype(np.float32)
    l_real, g_real = T.softmax_ce_mean(c_real, y_r)
    l_fake, g_fake = T.softmax_ce_mean(c_fake, y_f)
    l_unl, g_unl = T.c_unl_loss(c_unl, d_unl)
    l_ent, g_ent = T.entropy(c_unl)
    l_bal, g_bal = T.balance_entropy(c_unl)
    l_mse, g_mu, g_mr = T.mse_mean(c_unl, c_rep)
    if not with_rep:
        l_mse, g_mu, g_mr = 0.0, 0 * g_mu, 0 * g_mr
    ref = 0.005 * l_unl + l_real + 1e-6 * l_ent + 1e-3 * l_bal + lam1 * l_fake + lam2 * l_mse
    parts = [c_real, c_unl] + ([c_rep] if with_rep else []) + [c_fake]
    cl = np.zeros((sum(p.shape[0] for p in parts), 32), np.float32)
    cl[:, :10] = np.concatenate(parts)
    cld, dl, loss = dev(cl), torch.full(cl.shape, 7.0, device='cuda'), torch.zeros(1, device='cuda')
    yr, yf, du, lam = dev(y_r), dev(y_f), dev(d_unl), dev(np.array([lam1, lam2]))
    lib.call('tg_c_loss_f32', lib.ptr(cld), 32, n_r, n_u, n_u if with_rep else 0, n_f, lib.ptr(yr), lib.ptr(yf), lib.ptr(du), 1,
             lib.ptr(lam), lib.ptr(dl), 32, lib.ptr(loss), st())
    assert abs(loss.item() - ref) < 2e-5 * max(1, abs(ref))
    g = dl.cpu().numpy()
    gref = [g_real, 0.005 * g_unl + 1e-6 * g_ent + 1e-3 * g_bal + lam2 * g_mu] + ([lam2 * g_mr] if with_rep else []) + [lam1 * g_fake]
    gref = np.concatenate(gref)
    assert np.abs(g[:, :10] - gref).max() < 2e-4 * np.abs(gref).max()
    assert (g[:, 10:] == 0).all()


def test_feature_match_and_pull_away():
    lib = _lib()
    rng = np.random.default_rng(2)
    ff, fu = rng.standard_normal((20, 128)).astype(np.float32), rng.standard_normal((30, 128)).astype(np.float32)
    v, ga, gb = T.feature_match(ff, fu)
    a, b = dev(ff), dev(fu)
    da, db, loss = torch.zeros_like(a), torch.zeros_like(b), torch.zeros(1, device='cuda')
    lib.call('tg_feature_match_f32', lib.ptr(a), 20, lib.ptr(b), 30, 128, lib.ptr(da), lib.ptr(db), lib.ptr(loss), st())
    assert abs(loss.item() - v) < 1e-6
    np.testing.assert_allclose(da.cpu().numpy(), ga, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(db.cpu().numpy(), gb, rtol=1e-5, atol=1e-9)
    for masked, fn in ((1, T.pull_away_masked), (0, T.pull_away_unmasked)):
        v, g = fn(ff.astype(np.float64))
        scratch = torch.zeros(20 * 128 + 20 * 20 + 20, device='cuda')
        lib.call('tg_pull_away_f32', lib.ptr(a), 20, 128, masked, lib.ptr(scratch), lib.ptr(da), lib.ptr(loss), st())
        assert abs(loss.item() - v) < 1e-5 * max(1.0, abs(v))
        assert np.abs(da.cpu().numpy() - g).max() < 1e-4 * np.abs(g).max() + 1e-9


def test_adam_and_ema_bit_tight_on_identical_gradients():
    lib = _lib()
    rng = np.random.default_rng(3)
    n = 10007                                 # odd: exercises the scalar tail
    p0 = rng.standard_normal(n).astype(np.float32)
    p, m, v = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    lr = dev(np.array([3e-4]))
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    shadow = dev(p0)
    pr, mr, vr, sr = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32), p0.copy()
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        g[::7] *= 1e-9                         # gradients around Adam's epsilon
        g[::11] = 0
        gd = dev(2 * g)                        # grad_scale 0.5 = mean over 2 replicas
        lib.call('tg_adam_f32', lib.ptr(p), lib.ptr(gd), lib.ptr(m), lib.ptr(v), n, lib.ptr(lr), 0.5, 0.999, 1e-8, lib.ptr(step), 0.5, st())
        lib.call('tg_ema_f32', lib.ptr(shadow), lib.ptr(p), n, 0.9999, st())
        pr, mr, vr = T.adam_update(pr, g, mr, vr, t, 3e-4, 0.5)
        sr = T.ema_update(sr, pr)
    assert step.item() == 3
    np.testing.assert_allclose(m.cpu().numpy(), mr, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(v.cpu().numpy(), vr, rtol=1e-6, atol=1e-20)
    assert np.abs(p.cpu().numpy() - pr).max() <= 3e-4 * 3e-6 + 2e-7 * np.abs(pr).max()
    np.testing.assert_allclose(shadow.cpu().numpy(), sr, rtol=1e-6, atol=1e-7)


def test_argmax_onehot_and_accuracy():
    lib = _lib()
    rng = np.random.default_rng(4)
    lg = rng.standard_normal((77, 10)).astype(np.float32)
    lg[5, 3] = lg[5, 7] = 9.0                                   # tie: first index wins (tf.argmax)
    lp = np.zeros((77, 32), np.float32)
    lp[:, :10] = lg
    ld, out = dev(lp), torch.zeros(77 * 10, device='cuda')
    lib.call('tg_argmax_onehot_f32', lib.ptr(ld), 32, 77, 10, lib.ptr(out), st())
    np.testing.assert_array_equal(out.cpu().numpy().reshape(77, 10), T.argmax_onehot(lg))
    labels = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 77)]
    cnt, lab = torch.zeros(2, device='cuda'), dev(labels)
    for _ in range(2):
        lib.call('tg_accuracy_count_f32', lib.ptr(ld), 32, lib.ptr(lab), 77, 10, lib.ptr(cnt), st())
    correct = (lg.argmax(1) == labels.argmax(1)).sum()
    assert tuple(cnt.cpu().numpy()) == (2.0 * correct, 154.0)


def test_colstats_segments_ragged():
    lib = _lib()
    rng = np.random.default_rng(5)
    segs = [300, 7, 1025]
    rows, c, ld = sum(segs), 10, 32
    a = np.zeros((rows, ld), np.float32)
    a[:, :c] = rng.standard_normal((rows, c))
    y = np.zeros((rows, ld), np.float32)
    y[:, :c] = rng.standard_normal((rows, c))
    ad, yd = dev(a), dev(y)
    wsn = lib.call('tg_colstats_workspace_floats', rows, len(segs), c)
    work, s1, s2 = torch.zeros(wsn, device='cuda'), torch.zeros(3 * c, device='cuda'), torch.zeros(3 * c, device='cuda')
    sa = (C.c_int32 * 3)(*segs)
    off = np.cumsum([0] + segs)
    sl = [slice(off[i], off[i + 1]) for i in range(3)]
    lib.call('tg_colstats_f32', 1, lib.ptr(ad), ld, None, 0, rows, c, sa, 3, 0, 0.0, lib.ptr(work), lib.ptr(s1), lib.ptr(s2), st())
    np.testing.assert_allclose(s1.cpu().numpy().reshape(3, c), [a[s, :c].sum(0) for s in sl], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(s2.cpu().numpy().reshape(3, c), [(a[s, :c] ** 2).sum(0) for s in sl], rtol=1e-4, atol=1e-4)
    lib.call('tg_colstats_f32', 2, lib.ptr(ad), ld, lib.ptr(yd), ld, rows, c, sa, 3, 1, 0.2, lib.ptr(work), lib.ptr(s1), None, st())
    ref = [(a[s, :c] * np.where(y[s, :c] > 0, 1.0, 0.2)).sum(0) for s in sl]
    np.testing.assert_allclose(s1.cpu().numpy().reshape(3, c), ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(lib.TgError, match="segments sum"):
        lib.call('tg_colstats_f32', 0, lib.ptr(ad), ld, None, 0, rows + 1, c, sa, 3, 0, 0.0, lib.ptr(work), lib.ptr(s1), None, st())


def test_philox_streams_are_distinct_and_well_distributed():
    lib = _lib()
    n = 1 << 20
    state = torch.tensor([1234, 0], dtype=torch.int64, device='cuda')
    u, u2, mask, nrm = [torch.zeros(n, device='cuda') for _ in range(4)]
    lib.call('tg_rng_uniform_f32', lib.ptr(u), n, -1.0, 1.0, lib.ptr(state), 1, st())
    lib.call('tg_rng_uniform_f32', lib.ptr(u2), n, -1.0, 1.0, lib.ptr(state), 2, st())
    lib.call('tg_rng_keep_mask_f32', lib.ptr(mask), n, 0.8, lib.ptr(state), 3, st())
    lib.call('tg_rng_normal_f32', lib.ptr(nrm), n, 0.15, lib.ptr(state), 4, st())
    un, un2, mk, nm = u.cpu().numpy(), u2.cpu().numpy(), mask.cpu().numpy(), nrm.cpu().numpy()
    assert -1 <= un.min() and un.max() < 1 and abs(un.mean()) < 5e-3 and abs(un.var() - 1 / 3) < 5e-3
    assert abs(np.corrcoef(un, un2)[0, 1]) < 5e-3
    assert set(np.unique(mk)) == {0.0, 1.0} and abs(mk.mean() - 0.8) < 2e-3
    assert abs(nm.mean()) < 1e-3 and abs(nm.std() - 0.15) < 1e-3
    lib.call('tg_rng_advance', lib.ptr(state), st())           # a graph replay draws new numbers
    lib.call('tg_rng_uniform_f32', lib.ptr(u2), n, -1.0, 1.0, lib.ptr(state), 1, st())
    assert abs(np.corrcoef(un, u2.cpu().numpy())[0, 1]) < 5e-3
    oh = torch.zeros(1000 * 10, device='cuda')
    lib.call('tg_rng_onehot_f32', lib.ptr(oh), 1000, 10, lib.ptr(state), 5, st())
    o = oh.cpu().numpy().reshape(1000, 10)
    assert (o.sum(1) == 1).all() and o.sum(0).min() > 50


# 64-pixel images (aligned), 36-pixel ones (ragged: tiles straddle applications), and 16x16 / 32x32 ones: with bf16 operands those take
# the halo-tiled 3x3 kernel (csrc/conv3x3_bf16.hip: 256-pixel tiles of whole image rows)
@pytest.mark.parametrize("segs,h", [([2, 4, 1], 8), ([5, 5, 9], 6), ([2, 3, 1], 16), ([1, 2], 32)])
@pytest.mark.parametrize("prec", ['f32', 'bf16'])
def test_fused_mean_only_batch_norm_forward_backward(prec, segs, h):
    """tg_igemm_colsum_{f32,bf16} + tg_mobn_apply_f32 (training and evaluation) and tg_mobn_bwd_f32 against the oracle's
    conv2d + mean_only_batch_norm (Model/nn.py:147-187) applied per application segment, pop_mean updated sequentially."""
    from tg import geom
    lib = _lib()
    q = (lambda a: T.bf16_round(a)) if prec == 'bf16' else (lambda a: a)
    rng = np.random.default_rng(5)
    w, cin, cout = h, 64, 128
    n = sum(segs)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    pop0 = rng.standard_normal(cout).astype(np.float32)
    seg_rows = [s * h * w for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    # oracle
    y_ref, pop, o = [], pop0.astype(np.float64), 0
    pre = T.conv2d(q(x).astype(np.float64), q(wt).astype(np.float64))
    for s in segs:
        yy, pop = T.mobn_train(pre[o:o + s], pop, b.astype(np.float64))
        y_ref.append(T.lrelu(yy, 0.2))
        o += s
    y_ref = np.concatenate(y_ref)
    # HIP
    w_oti = np.zeros((cout, 9, cin), np.float32)
    w_oti[:] = wt.reshape(9, cin, cout).transpose(2, 0, 1)
    xd, wd, bd, popd = dev(x), dev(w_oti), dev(b), dev(pop0)
    yd = torch.full((n, h, w, cout), 7.0, device='cuda')
    sums = torch.zeros(16 * len(segs) * cout, device='cuda')         # room for the 8 accumulator replicas of tg_mobn_bwd_f32
    d = geom.conv_fwd(n, h, w, cin, cout, 3, 1, 'SAME')
    was, halo0 = lib.call('tg_conv3x3_policy', 1), lib.call('tg_conv3x3_launches')        # the halo kernel wherever the layer applies
    lib.call_igemm('tg_igemm_colsum_' + prec, d, lib.ptr(xd), lib.ptr(wd), lib.ptr(yd), sa, len(segs), lib.ptr(sums), 0, st())
    lib.call('tg_conv3x3_policy', was)
    assert lib.call('tg_conv3x3_launches') - halo0 == int(h in (16, 32))
    pre_hip = yd.cpu().numpy().copy()
    # element-wise bound from the ACTUAL sum |a||b| behind each output (tests/test_gpu_igemm.py docstring: 1e-6 of it covers fp32 accumulation in
    # any order with a factor five to spare and rejects bf16-rounded operands by two orders of magnitude)
    pre_abs = T.conv2d(np.abs(q(x).astype(np.float64)), np.abs(q(wt).astype(np.float64)))
    scale = np.abs(x).max() * np.abs(wt).max() * 9 * cin
    assert (np.abs(pre_hip - pre) <= 1e-6 * pre_abs).all(), float((np.abs(pre_hip - pre) / pre_abs).max())
    lib.call('tg_mobn_apply_f32', lib.ptr(yd), cout, n * h * w, cout, sa, len(segs), lib.ptr(sums), lib.ptr(bd), lib.ptr(popd), 0.9,
             lib.ACT['lrelu'], 0.2, st())
    # after the mean-only BN: the output's own bound + the same bound on the segment mean it subtracts + the shift b
    seg_mean_abs = np.concatenate([np.broadcast_to(pre_abs[o:o + s_].mean(axis=(0, 1, 2)), pre_abs[o:o + s_].shape)
                                   for o, s_ in zip(np.cumsum([0] + list(segs[:-1])), segs)])
    assert (np.abs(yd.cpu().numpy() - y_ref) <= 1e-6 * (pre_abs + seg_mean_abs + np.abs(b))).all()
    np.testing.assert_allclose(popd.cpu().numpy(), pop, rtol=2e-5, atol=2e-6)
    # evaluation mode: sums = NULL -> x - pop_mean + b
    ye = dev(pre_hip)
    lib.call('tg_mobn_apply_f32', lib.ptr(ye), cout, n * h * w, cout, sa, len(segs), None, lib.ptr(bd), lib.ptr(popd), 0.9,
             lib.ACT['lrelu'], 0.2, st())
    np.testing.assert_allclose(ye.cpu().numpy(), T.lrelu(pre_hip - popd.cpu().numpy() + b, 0.2), rtol=1e-6, atol=1e-6)
    # backward: dpre = dy*lrelu'(y) - mean_seg(...), db = sum over all rows
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    y_act = yd.cpu().numpy()
    t = T.lrelu_bwd_from_out(y_act.astype(np.float64), dy.astype(np.float64), 0.2)
    dx_ref, o = [], 0
    for s in segs:
        dxs, _ = T.mobn_train_bwd(t[o:o + s])
        dx_ref.append(dxs)
        o += s
    dx_ref, db_ref = np.concatenate(dx_ref), t.sum(axis=(0, 1, 2))
    dyd, dxd, dbd = dev(dy), torch.full((n, h, w, cout), 7.0, device='cuda'), torch.full((cout,), 7.0, device='cuda')
    lib.call('tg_mobn_bwd_f32', lib.ptr(dyd), cout, lib.ptr(yd), cout, lib.ptr(dxd), cout, n * h * w, cout, sa, len(segs), lib.ACT['lrelu'], 0.2,
             lib.ptr(sums), 0, lib.ptr(dbd), st())
    np.testing.assert_allclose(dxd.cpu().numpy(), dx_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dbd.cpu().numpy(), db_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("n_split,t,c_pad,n_pad,c,n", [(512, 1, 32, 128, 27, 128), (56, 9, 128, 128, 128, 128), (3, 25, 64, 32, 42, 3), (40, 1, 32, 32, 5, 7)])
def test_slab_reduce_both_paths(n_split, t, c_pad, n_pad, c, n):
    """tg_slab_reduce_f32: dst[t][c][n] = sum over the pixel-split slabs, padding dropped; the few-outputs / many-slabs path
    (first convolution, 512 splits) and the one-thread-per-output path; fixed summation order -> identical on repetition."""
    lib = _lib()
    rng = np.random.default_rng(11)
    slab = rng.standard_normal((n_split, t, c_pad, n_pad)).astype(np.float32)
    sd = dev(slab)
    out = [torch.full((t * c * n,), 7.0, device='cuda') for _ in range(2)]
    for o in out:
        lib.call('tg_slab_reduce_f32', lib.ptr(sd), n_split, t, c_pad, n_pad, c, n, lib.ptr(o), st())
    ref = slab[:, :, :c, :n].astype(np.float64).sum(0).reshape(-1)
    got = out[0].cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(slab).sum(0).max()
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("segs,hw,c,ld", [([3, 5], 6, 128, 128), ([50, 50, 100], 1, 10, 32), ([7], 4, 300, 320), ([2, 1, 4], 8, 64, 64)])
def test_fused_segmented_batch_norm(segs, hw, c, ld):
    """tg_bn_train_f32 / tg_bn_train_bwd_f32 against the oracle's batch_norm_train(+_bwd) applied per application segment
    (Model/modle_base.py:229-237): ragged segments, a channel count that is not a multiple of 4 (the [N,10] logits), more than
    one 256-column block; moving statistics updated segment by segment with the Bessel-corrected variance."""
    lib = _lib()
    rng = np.random.default_rng(3)
    n = sum(segs)
    x = (rng.standard_normal((n, hw, hw, c)) * 2 + 0.7).astype(np.float32)
    gamma, beta = rng.standard_normal(c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    mm0, mv0 = rng.standard_normal(c).astype(np.float32), (rng.random(c) + 0.5).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    seg_rows = [s * hw * hw for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    y_ref, dx_ref, dg_ref, db_ref = [], [], np.zeros(c), np.zeros(c)
    mm, mv, o = mm0.astype(np.float64), mv0.astype(np.float64), 0
    for s in segs:
        xs = x[o:o + s].astype(np.float64)
        ys, cache = T.batch_norm_train(xs, gamma.astype(np.float64), beta.astype(np.float64), 1e-5)
        mm, mv = T.batch_norm_moving_update(mm, mv, cache[2], cache[3], s * hw * hw, 0.9, True)
        dxs, dg, db = T.batch_norm_train_bwd(dy[o:o + s].astype(np.float64), gamma.astype(np.float64), cache)
        y_ref.append(ys); dx_ref.append(dxs); dg_ref += dg; db_ref += db
        o += s
    y_ref, dx_ref = np.concatenate(y_ref), np.concatenate(dx_ref)

    def pad(a):
        out = np.zeros(a.shape[:-1] + (ld,), np.float32)
        out[..., :c] = a
        return out
    xd, dyd = dev(pad(x)), dev(pad(dy))
    gd, bd, mmd, mvd = dev(gamma), dev(beta), dev(mm0), dev(mv0)
    yd, dxd = torch.full((n, hw, hw, ld), 7.0, device='cuda'), torch.full((n, hw, hw, ld), 7.0, device='cuda')
    sums = torch.zeros(32 * len(segs) * c, device='cuda')
    mean_inv = torch.zeros(2 * len(segs) * c, device='cuda')
    dgd, dbd = torch.full((c,), 7.0, device='cuda'), torch.full((c,), 7.0, device='cuda')
    rows = n * hw * hw
    lib.call('tg_bn_train_f32', lib.ptr(xd), ld, lib.ptr(yd), ld, rows, c, sa, len(segs), lib.ptr(gd), lib.ptr(bd), 1e-5, 0.9, lib.ptr(mmd), lib.ptr(mvd),
             lib.ptr(sums), 0, lib.ptr(mean_inv), st())
    lib.call('tg_bn_train_bwd_f32', lib.ptr(dyd), ld, lib.ptr(xd), ld, lib.ptr(dxd), ld, rows, c, sa, len(segs), lib.ptr(gd), lib.ptr(mean_inv), 0,
             lib.ptr(sums), 0, lib.ptr(dgd), lib.ptr(dbd), st())
    y = yd.cpu().numpy()
    np.testing.assert_allclose(y[..., :c], y_ref, rtol=2e-5, atol=2e-5)
    cp = (c + 3) // 4 * 4
    assert (y[..., c:cp] == 0).all()                                  # columns up to the next multiple of 4 are written as zeros
    np.testing.assert_allclose(dxd.cpu().numpy()[..., :c], dx_ref, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dgd.cpu().numpy(), dg_ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dbd.cpu().numpy(), db_ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(mmd.cpu().numpy(), mm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mvd.cpu().numpy(), mv, rtol=1e-4, atol=1e-6)
    # round 3: the same pass with the derivative of the activation that produced x (conv -> leaky relu -> BN) and that layer's bias gradient
    # folded in (tg_bn_train_bwd_act_f32): dx * act'(x), bias gradient = its column sums; where the column layout allows the sums
    if c % 4 == 0 and (c <= 256 and 256 % (c // 4) == 0 or c % 256 == 0):
        for act, alpha in (('lrelu', 0.2), ('relu', 0.0), (None, 0.0)):
            mult = np.where(x > 0, 1.0, alpha) if act else np.ones_like(x, np.float64)
            ref = dx_ref * mult
            dxa = torch.full((n, hw, hw, ld), 7.0, device='cuda')
            dsum = torch.zeros(16 * c, device='cuda')
            dbias = torch.full((c,), 7.0, device='cuda')
            dgd.fill_(7.0); dbd.fill_(7.0)
            lib.call('tg_bn_train_bwd_act_f32', lib.ptr(dyd), ld, lib.ptr(xd), ld, lib.ptr(dxa), ld, rows, c, sa, len(segs), lib.ptr(gd), lib.ptr(mean_inv),
                     lib.ACT[act], alpha, lib.ptr(sums), 0, lib.ptr(dgd), lib.ptr(dbd), lib.ptr(dsum), 0, lib.ptr(dbias), st())
            np.testing.assert_allclose(dxa.cpu().numpy()[..., :c], ref, rtol=2e-4, atol=2e-5)
            np.testing.assert_allclose(dbias.cpu().numpy(), ref.reshape(-1, c).sum(0), rtol=2e-4, atol=2e-3)
            np.testing.assert_allclose(dgd.cpu().numpy(), dg_ref, rtol=2e-4, atol=2e-4)        # the batch norm's own gradients are unchanged
            np.testing.assert_allclose(dbd.cpu().numpy(), db_ref, rtol=2e-4, atol=2e-4)
    else:
        with pytest.raises(lib.TgError, match='bias-gradient sums'):
            lib.call('tg_bn_train_bwd_act_f32', lib.ptr(dyd), ld, lib.ptr(xd), ld, lib.ptr(dxd), ld, rows, c, sa, len(segs), lib.ptr(gd), lib.ptr(mean_inv),
                     lib.ACT['lrelu'], 0.2, lib.ptr(sums), 0, None, None, lib.ptr(sums), 0, lib.ptr(dbd), st())


@pytest.mark.parametrize("rows,c,ld_out,act", [(1000, 32, 32, 'lrelu'), (77, 3, 32, 'tanh'), (5000, 138, 160, None), (64, 1, 32, None)])
def test_actgrad_with_bias_gradient(rows, c, ld_out, act):
    """tg_actgrad_bias_f32: dpre = dy*act'(y) with zeroed padding and bias_grad = column sums of dpre, in one pass."""
    lib = _lib()
    rng = np.random.default_rng(2)
    dy = rng.standard_normal((rows, c)).astype(np.float32)
    y = np.tanh(rng.standard_normal((rows, c))).astype(np.float32)
    if act == 'lrelu':
        ref = T.lrelu_bwd_from_out(y.astype(np.float64), dy.astype(np.float64), 0.2)
    elif act == 'tanh':
        ref = dy.astype(np.float64) * (1 - y.astype(np.float64) ** 2)
    else:
        ref = dy.astype(np.float64)
    dyd, ydv = dev(dy), dev(y)
    out = torch.full((rows, ld_out), 7.0, device='cuda')
    sums = torch.zeros(16 * c, device='cuda')
    bg = torch.full((c,), 7.0, device='cuda')
    lib.call('tg_actgrad_bias_f32', lib.ptr(dyd), c, lib.ptr(ydv) if act else None, c, lib.ptr(out), ld_out, rows, c, lib.ACT[act], 0.2,
             lib.ptr(sums), 0, lib.ptr(bg), st())
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[:, :c], ref, rtol=1e-6, atol=1e-6)
    assert (o[:, c:] == 0).all()
    np.testing.assert_allclose(bg.cpu().numpy(), ref.sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
@pytest.mark.parametrize("segs,h,cin,cout", [([2, 4, 1], 8, 64, 96), ([5, 5, 9], 6, 64, 96), ([2, 1, 3], 16, 128, 256), ([1, 2], 32, 128, 128)])
def test_input_gradient_fused_with_mobn_backward_statistics(prec, segs, h, cin, cout):
    """tg_igemm_actsum_* (input gradient of a conv, times act'(y) of the mean-only-BN layer that produced its input, with per-application
    column sums) + tg_mobn_center_f32 == conv2d_bwd_input -> lrelu' -> mean_only_batch_norm backward of the oracle."""
    from tg import geom
    lib = _lib()
    q = (lambda a: T.bf16_round(a)) if prec == 'bf16' else (lambda a: a)
    rng = np.random.default_rng(9)
    n, w_ = sum(segs), h                              # the differentiated conv: cin -> cout; its input is the MOBN layer's output y
    y = rng.standard_normal((n, h, w_, cin)).astype(np.float32)          # activated output of the producing layer (sign decides lrelu')
    wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    dpre = rng.standard_normal((n, h, w_, cout)).astype(np.float32)      # gradient at the conv's pre-activation
    seg_rows = [s * h * w_ for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    gx = T.conv2d_bwd_input(y.shape, q(wt).astype(np.float64), q(dpre).astype(np.float64))
    t_ref = T.lrelu_bwd_from_out(y.astype(np.float64), gx, 0.2)
    dx_ref, o = [], 0
    for s in segs:
        dxs, _ = T.mobn_train_bwd(t_ref[o:o + s])
        dx_ref.append(dxs)
        o += s
    dx_ref, db_ref = np.concatenate(dx_ref), t_ref.sum(axis=(0, 1, 2))
    co_p = geom.pad32(cout)
    w_hwio = np.zeros((9, cin, co_p), np.float32)
    w_hwio[:, :, :cout] = wt.reshape(9, cin, cout)
    dp = np.zeros((n, h, w_, co_p), np.float32)
    dp[..., :cout] = dpre
    yd, wd, dd = dev(y), dev(w_hwio), dev(dp)
    td = torch.full((n, h, w_, cin), 7.0, device='cuda')
    sums = torch.zeros(2 * len(segs) * cin, device='cuda')
    descs = geom.conv_dgrad(n, h, w_, cin, co_p, 3, 1, 'SAME')
    assert len(descs) == 1
    was, halo0 = lib.call('tg_conv3x3_policy', 1), lib.call('tg_conv3x3_launches')        # the halo kernel wherever the layer applies
    lib.call_igemm('tg_igemm_actsum_' + prec, descs[0], lib.ptr(dd), lib.ptr(wd), lib.ptr(yd), lib.ACT['lrelu'], 0.2, lib.ptr(td), sa, len(segs),
             lib.ptr(sums), 0, st())
    lib.call('tg_conv3x3_policy', was)
    assert lib.call('tg_conv3x3_launches') - halo0 == int(h in (16, 32))
    scale = np.abs(dpre).max() * np.abs(wt).max() * 9 * cout
    gx_abs = T.conv2d_bwd_input(y.shape, np.abs(q(wt).astype(np.float64)), np.abs(q(dpre).astype(np.float64)))
    assert (np.abs(td.cpu().numpy() - t_ref) <= 1e-6 * gx_abs).all(), float((np.abs(td.cpu().numpy() - t_ref) / gx_abs).max())
    dxd, dbd = torch.full((n, h, w_, cin), 7.0, device='cuda'), torch.full((cin,), 7.0, device='cuda')
    lib.call('tg_mobn_center_f32', lib.ptr(td), cin, lib.ptr(dxd), cin, n * h * w_, cin, sa, len(segs), lib.ptr(sums), 1, lib.ptr(dbd), st())
    assert np.abs(dxd.cpu().numpy() - dx_ref).max() <= 3e-5 * scale
    np.testing.assert_allclose(dbd.cpu().numpy(), db_ref, rtol=1e-4, atol=3e-5 * scale * 8)


def test_maxpool_backward_fused_with_mobn_backward_statistics():
    """tg_maxpool2_bwd_actsum_f32 + tg_mobn_center_f32 (8 accumulator replicas) == max-pool/dropout backward -> lrelu' -> mean-only-BN
    backward of the oracle, per application segment."""
    lib = _lib()
    rng = np.random.default_rng(12)
    segs, h, c = [3, 1, 2], 8, 64
    n = sum(segs)
    y = rng.standard_normal((n, h, h, c)).astype(np.float32)
    dpool = rng.standard_normal((n, h // 2, h // 2, c)).astype(np.float32)
    mask = (rng.random(dpool.shape) < 0.5).astype(np.float32)
    _, idx = T.maxpool2(y.astype(np.float64))
    gy = T.maxpool2_bwd((dpool * mask * 2.0).astype(np.float64), idx, y.shape)
    t_ref = T.lrelu_bwd_from_out(y.astype(np.float64), gy, 0.2)
    dx_ref, o = [], 0
    for s in segs:
        dxs, _ = T.mobn_train_bwd(t_ref[o:o + s])
        dx_ref.append(dxs)
        o += s
    dx_ref = np.concatenate(dx_ref)
    seg_rows = [s * h * h for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    yd, dd, md = dev(y), dev(dpool), dev(mask)
    td = torch.full((n, h, h, c), 7.0, device='cuda')
    sums = torch.zeros(16 * len(segs) * c, device='cuda')
    lib.call('tg_maxpool2_bwd_actsum_f32', lib.ptr(dd), c, lib.ptr(md), c, 2.0, lib.ptr(yd), c, lib.ptr(td), c, n, h, h, c, sa, len(segs),
             lib.ACT['lrelu'], 0.2, lib.ptr(sums), 0, st())
    np.testing.assert_allclose(td.cpu().numpy(), t_ref, rtol=1e-6, atol=1e-6)
    dxd, dbd = torch.full((n, h, h, c), 7.0, device='cuda'), torch.full((c,), 7.0, device='cuda')
    lib.call('tg_mobn_center_f32', lib.ptr(td), c, lib.ptr(dxd), c, n * h * h, c, sa, len(segs), lib.ptr(sums), 8, lib.ptr(dbd), st())
    np.testing.assert_allclose(dxd.cpu().numpy(), dx_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dbd.cpu().numpy(), t_ref.sum(axis=(0, 1, 2)), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
@pytest.mark.parametrize("hw,cin,cout,n,segs", [(16, 128, 128, 300, [100, 200]), (32, 64, 256, 150, [150])])
def test_halo_kernel_walks_several_tiles_per_workgroup(prec, hw, cin, cout, n, segs):
    """csrc/conv3x3_bf16.hip, persistent tile-pipelined form: with more tiles than compute units a workgroup walks several tiles (operands of
    the next tile fetched during the current one, column sums handed to the loader waves tile by tile).  300 images of 16x16 are 300 tiles,
    150 of 32x32 with 256 output channels 1 200.  Checked against the generic implicit GEMM of the same library (itself checked against the
    oracle above and in test_gpu_igemm.py) on the forward pass with bias + leaky relu, the forward pass with per-application column sums,
    and the input gradient with the activation-gradient multiplier and column sums; tolerance = fp32 accumulation order."""
    from tg import geom
    lib = _lib()
    rng = np.random.default_rng(11)
    x = dev(rng.standard_normal((n, hw, hw, cin)))
    w_oti = dev(rng.standard_normal((cout, 9, cin)) * 0.1)
    w_hwio = dev(rng.standard_normal((9, cin, cout)) * 0.1)
    bias = dev(rng.standard_normal(cout))
    dy = dev(rng.standard_normal((n, hw, hw, cout)))
    sa = (C.c_int32 * len(segs))(*[s * hw * hw for s in segs])
    d_act, d_lin = geom.conv_fwd(n, hw, hw, cin, cout, 3, 1, 'SAME', act='lrelu'), geom.conv_fwd(n, hw, hw, cin, cout, 3, 1, 'SAME')
    d_bwd = geom.conv_dgrad(n, hw, hw, cin, cout, 3, 1, 'SAME')[0]
    outs = {}
    for policy in (1, 2):                             # 1: the halo kernel wherever it applies, 2: never
        was, halo0 = lib.call('tg_conv3x3_policy', policy), lib.call('tg_conv3x3_launches')
        y1 = torch.full((n, hw, hw, cout), 7.0, device='cuda')
        lib.call_igemm('tg_igemm_' + prec, d_act, lib.ptr(x), lib.ptr(w_oti), lib.ptr(bias), lib.ptr(y1), st())
        y2, s2 = torch.full((n, hw, hw, cout), 7.0, device='cuda'), torch.zeros(2 * len(segs) * cout, device='cuda')
        lib.call_igemm('tg_igemm_colsum_' + prec, d_lin, lib.ptr(x), lib.ptr(w_oti), lib.ptr(y2), sa, len(segs), lib.ptr(s2), 0, st())
        g3, s3 = torch.full((n, hw, hw, cin), 7.0, device='cuda'), torch.zeros(2 * len(segs) * cin, device='cuda')
        lib.call_igemm('tg_igemm_actsum_' + prec, d_bwd, lib.ptr(dy), lib.ptr(w_hwio), lib.ptr(x), lib.ACT['lrelu'], 0.2, lib.ptr(g3), sa, len(segs),
                 lib.ptr(s3), 0, st())                # the halo kernel needs 128 | output channels: the input gradient of the second case stays generic
        lib.call('tg_conv3x3_policy', was)
        assert lib.call('tg_conv3x3_launches') - halo0 == (2 + int(cin % 128 == 0) if policy == 1 else 0)
        outs[policy] = [t.cpu().numpy() for t in (y1, y2, s2, g3, s3)]
    scale = float(x.abs().max()) * 0.5 * 9 * cin
    scale_g = float(dy.abs().max()) * 0.5 * 9 * cout
    for i, sc in ((0, scale), (1, scale), (3, scale_g)):
        assert np.abs(outs[1][i] - outs[2][i]).max() <= 3e-5 * sc, i
    sums = lambda a: np.frombuffer(a.tobytes(), np.float64)              # the column sums are doubles in a float32 tensor's storage
    np.testing.assert_allclose(sums(outs[1][2]), sums(outs[2][2]), rtol=1e-5, atol=1e-6 * scale * segs[-1] * hw * hw)
    np.testing.assert_allclose(sums(outs[1][4]), sums(outs[2][4]), rtol=1e-5, atol=1e-6 * scale_g * segs[-1] * hw * hw)


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
@pytest.mark.parametrize("hw,cin,cout,n", [(32, 128, 128, 40), (16, 128, 256, 40), (16, 256, 256, 250), (32, 128, 128, 130)])
def test_filter_gradient_with_the_activation_tile_read_once_equals_the_generic_kernel(prec, hw, cin, cout, n):
    """csrc/wgrad3x3.hip under the default routing with the library's own pixel split (tg_wgrad_splits[_bf16]: one workgroup per CU, 10 - 62
    tiles each, two tiles of loads in flight) against the generic per-tap kernel with ITS split (tg_conv3x3_policy 2), both reduced over their
    slabs in float64: the launches of the long-horizon run (40 images) and of the bench line (130 / 250)."""
    from tg import geom
    lib = _lib()
    rng = np.random.default_rng(13)
    x = dev(rng.standard_normal((n, hw, hw, cin)))
    dy = dev(rng.standard_normal((n, hw, hw, cout)))
    d = geom.conv_wgrad(n, hw, hw, cin, cout, 3, 1, 'SAME')
    got = {}
    for policy in (0, 2):
        was, halo0 = lib.call('tg_conv3x3_policy', policy), lib.call('tg_conv3x3_launches')
        ns = geom.wgrad_splits(d, prec == 'bf16')
        slab = torch.full((ns, 9, cin, cout), 7.0, device='cuda')
        lib.call('tg_wgrad_' + prec, d, lib.ptr(x), lib.ptr(dy), lib.ptr(slab), ns, st())
        lib.call('tg_conv3x3_policy', was)
        assert lib.call('tg_conv3x3_launches') - halo0 == (1 if policy == 0 else 0), (policy, ns)
        got[policy] = (ns, slab.cpu().numpy().astype(np.float64).sum(0))
    assert got[0][0] * (cin // 32) * (cout // 128) in (256, 2 * 256 // 2) and got[0][0] != got[2][0]
    scale = float(x.abs().max()) * float(dy.abs().max()) * np.sqrt(n * hw * hw)          # random-sign sums grow like sqrt(pixels)
    assert np.abs(got[0][1] - got[2][1]).max() <= (2e-2 if prec == 'bf16' else 2e-5) * scale * 4


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
@pytest.mark.parametrize("segs", [[50, 80], [129, 1], [130]])
def test_halo_kernel_takes_the_whole_rounds_of_a_launch_and_the_generic_kernel_the_rest(prec, segs):
    """Default routing (tg_conv3x3_policy 0): 130 images of 32x32x128 -> 128 are 520 halo tiles = 2.03 rounds of one workgroup per CU; the
    launch is split — 128 leading images (512 tiles) on the halo kernel, 2 on the generic implicit GEMM — over disjoint image ranges of the
    same buffers, the column sums of both parts landing in the same per-application accumulators, also when an application boundary falls
    inside the head ([50, 80]) or inside the tail ([129, 1]).  Compared with the unsplit generic launch (policy 2)."""
    from tg import geom
    lib = _lib()
    n, hw, cin, cout = sum(segs), 32, 128, 128
    rng = np.random.default_rng(12)
    x = dev(rng.standard_normal((n, hw, hw, cin)))
    w_oti = dev(rng.standard_normal((cout, 9, cin)) * 0.1)
    w_hwio = dev(rng.standard_normal((9, cin, cout)) * 0.1)
    bias = dev(rng.standard_normal(cout))
    dy = dev(rng.standard_normal((n, hw, hw, cout)))
    sa = (C.c_int32 * len(segs))(*[s * hw * hw for s in segs])
    d_act, d_lin = geom.conv_fwd(n, hw, hw, cin, cout, 3, 1, 'SAME', act='lrelu'), geom.conv_fwd(n, hw, hw, cin, cout, 3, 1, 'SAME')
    d_bwd = geom.conv_dgrad(n, hw, hw, cin, cout, 3, 1, 'SAME')[0]
    outs = {}
    for policy in (0, 2):
        was, halo0 = lib.call('tg_conv3x3_policy', policy), lib.call('tg_conv3x3_launches')
        y1 = torch.full((n, hw, hw, cout), 7.0, device='cuda')
        lib.call_igemm('tg_igemm_' + prec, d_act, lib.ptr(x), lib.ptr(w_oti), lib.ptr(bias), lib.ptr(y1), st())
        y2, s2 = torch.full((n, hw, hw, cout), 7.0, device='cuda'), torch.zeros(2 * len(segs) * cout, device='cuda')
        lib.call_igemm('tg_igemm_colsum_' + prec, d_lin, lib.ptr(x), lib.ptr(w_oti), lib.ptr(y2), sa, len(segs), lib.ptr(s2), 0, st())
        g3, s3 = torch.full((n, hw, hw, cin), 7.0, device='cuda'), torch.zeros(2 * len(segs) * cin, device='cuda')
        lib.call_igemm('tg_igemm_actsum_' + prec, d_bwd, lib.ptr(dy), lib.ptr(w_hwio), lib.ptr(x), lib.ACT['lrelu'], 0.2, lib.ptr(g3), sa, len(segs),
                 lib.ptr(s3), 0, st())
        lib.call('tg_conv3x3_policy', was)
        assert lib.call('tg_conv3x3_launches') - halo0 == (3 if policy == 0 else 0)
        outs[policy] = [t.cpu().numpy() for t in (y1, y2, s2, g3, s3)]
    scale, scale_g = float(x.abs().max()) * 0.5 * 9 * cin, float(dy.abs().max()) * 0.5 * 9 * cout
    for i, sc in ((0, scale), (1, scale), (3, scale_g)):
        assert np.abs(outs[0][i] - outs[2][i]).max() <= 3e-5 * sc, i
    sums = lambda a: np.frombuffer(a.tobytes(), np.float64)
    np.testing.assert_allclose(sums(outs[0][2]), sums(outs[2][2]), rtol=1e-5, atol=1e-6 * scale * max(segs) * hw * hw)
    np.testing.assert_allclose(sums(outs[0][4]), sums(outs[2][4]), rtol=1e-5, atol=1e-6 * scale_g * max(segs) * hw * hw)


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
@pytest.mark.parametrize("n,hw,ci,co,segs,act", [(6, 16, 128, 128, [2, 4], 'lrelu'),      # the halo-tiled kernel (policy 1)
                                                 (5, 8, 256, 512, [3, 2], 'lrelu'),        # generic kernel, 8x8 images (cut tiles: fix-up epilogue)
                                                 (7, 6, 64, 96, [3, 4], 'relu'),           # application boundary inside a tile, overhanging rows
                                                 (4, 8, 64, 64, [4], None)])
def test_batch_norm_statistics_in_the_convolution_epilogue(prec, n, hw, ci, co, segs, act):
    """tg_igemm_bnstat_* + tg_bn_train_apply_f32 against tg_igemm_* + tg_bn_train_f32 (conv -> leaky relu -> batch norm of the SVHN / MNIST
    classifier, Model/Good_GAN.py:249-350): same stored activation, same normalised output, same moving statistics."""
    lib = _lib()
    from tg import geom
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.standard_normal((n, hw, hw, ci)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((co, 9, ci)) * 0.05).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).cuda()
    gamma, beta = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).cuda(), torch.from_numpy(rng.standard_normal(co).astype(np.float32)).cuda()
    d = geom.conv_fwd(n, hw, hw, ci, co, 3, 1, 'SAME', act=act)
    rows = n * hw * hw
    seg_rows = [s * hw * hw for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    was = lib.call('tg_conv3x3_policy', 1)
    try:
        out = {}
        for mode in ('plain', 'fused'):
            y = torch.full((n, hw, hw, co), 7.0, device='cuda')
            z = torch.full((n, hw, hw, co), 7.0, device='cuda')
            sums = torch.full((32 * len(segs) * co,), 7.0, device='cuda')          # garbage: both paths clear what they use
            mi = torch.zeros(2 * len(segs) * co, device='cuda')
            mm, mv = torch.zeros(co, device='cuda'), torch.ones(co, device='cuda')
            if mode == 'plain':
                lib.call_igemm('tg_igemm_' + prec, d, lib.ptr(x), lib.ptr(w), lib.ptr(bias), lib.ptr(y), st())
                lib.call('tg_bn_train_f32', lib.ptr(y), co, lib.ptr(z), co, rows, co, sa, len(segs), lib.ptr(gamma), lib.ptr(beta), 1e-5, 0.9, lib.ptr(mm),
                         lib.ptr(mv), lib.ptr(sums), 0, lib.ptr(mi), st())
            else:
                lib.call_igemm('tg_igemm_bnstat_' + prec, d, lib.ptr(x), lib.ptr(w), lib.ptr(bias), lib.ptr(y), sa, len(segs), lib.ptr(sums), 0, st())
                lib.call('tg_bn_train_apply_f32', lib.ptr(y), co, lib.ptr(z), co, rows, co, sa, len(segs), lib.ptr(gamma), lib.ptr(beta), 1e-5, 0.9,
                         lib.ptr(mm), lib.ptr(mv), lib.ptr(sums), lib.ptr(mi), st())
            torch.cuda.synchronize()
            out[mode] = [t.cpu().numpy().astype(np.float64) for t in (y, z, mi, mm, mv)]
    finally:
        lib.call('tg_conv3x3_policy', was)
    scale = float(x.abs().max() * w.abs().max()) * 9 * ci
    names = ('conv output', 'normalised output', 'mean / inv-std', 'moving mean', 'moving variance')
    for nm, a, b in zip(names, out['fused'], out['plain']):
        tol = 3e-5 * scale if nm == 'conv output' else 2e-4 * max(1.0, np.abs(b).max())
        assert np.abs(a - b).max() <= tol, (nm, np.abs(a - b).max(), tol)
    assert not (out['fused'][1] == 7.0).any()


@pytest.mark.parametrize("prec", ['f32', 'bf16'])
@pytest.mark.parametrize("n,hw,ci,co,segs,act", [(6, 16, 128, 128, [2, 4], 'lrelu'),      # the halo-tiled kernel (policy 1)
                                                 (5, 8, 256, 512, [3, 2], 'lrelu'),        # generic kernel, 8x8 images (cut tiles: fix-up epilogue)
                                                 (7, 6, 64, 96, [3, 4], 'relu'),           # application boundary inside a tile, overhanging rows
                                                 (9, 16, 128, 256, [9], None)])            # 128-wide tiles of the generic kernel
def test_batch_norm_backward_statistics_in_the_epilogue_of_the_launch_that_produces_dy(prec, n, hw, ci, co, segs, act):
    """tg_igemm_bnbwdstat_* + tg_bn_train_bwd_act_f32(sums_zeroed = 2) against tg_igemm_* + tg_bn_train_bwd_act_f32: the launch that
    writes the gradient dy of a batch norm's output (the next convolution's input gradient: any conv-shaped launch here) also takes
    sum dy and sum dy * x, and the backward pass that follows skips its statistics launch: same dy, same dx, same dgamma / dbeta / dbias."""
    lib = _lib()
    from tg import geom
    rng = np.random.default_rng(22)
    g_in = torch.from_numpy(rng.standard_normal((n, hw, hw, ci)).astype(np.float32)).cuda()          # what the launch convolves (dpre of the next layer)
    w = torch.from_numpy((rng.standard_normal((co, 9, ci)) * 0.05).astype(np.float32)).cuda()
    xbn = torch.from_numpy(rng.standard_normal((n, hw, hw, co)).astype(np.float32)).cuda()          # the batch norm's input (an activation's output)
    gamma = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).cuda()
    d = geom.conv_fwd(n, hw, hw, ci, co, 3, 1, 'SAME')
    rows = n * hw * hw
    seg_rows = [s * hw * hw for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    # mean / inv-std per (segment, channel) as the forward pass leaves them
    xs = xbn.double().reshape(rows, co)
    mi = torch.empty(len(segs), 2, co, dtype=torch.float64)
    r0 = 0
    for i, r in enumerate(seg_rows):
        blk = xs[r0:r0 + r]
        mi[i, 0] = blk.mean(0).cpu()
        mi[i, 1] = (1.0 / torch.sqrt(blk.var(0, unbiased=False) + 1e-5)).cpu()
        r0 += r
    mean_inv = mi.float().reshape(-1).cuda()
    was = lib.call('tg_conv3x3_policy', 1)
    try:
        out = {}
        for mode in ('plain', 'fused'):
            dy = torch.full((n, hw, hw, co), 7.0, device='cuda')
            dx = torch.full((n, hw, hw, co), 7.0, device='cuda')
            sums = torch.full((32 * len(segs) * co,), 7.0, device='cuda')          # garbage: both paths clear what they use
            dsum = torch.zeros(16 * co, device='cuda')
            dgamma, dbeta, dbias = (torch.zeros(co, device='cuda') for _ in range(3))
            if mode == 'plain':
                lib.call_igemm('tg_igemm_' + prec, d, lib.ptr(g_in), lib.ptr(w), None, lib.ptr(dy), st())
                flag = 0
            else:
                lib.call_igemm('tg_igemm_bnbwdstat_' + prec, d, lib.ptr(g_in), lib.ptr(w), lib.ptr(xbn), lib.ptr(dy), sa, len(segs), lib.ptr(sums), 0, st())
                flag = 2
            with_bias = co <= 256 and 256 % (co // 4) == 0 or co % 256 == 0        # column layouts the fused bias-gradient sums take
            lib.call('tg_bn_train_bwd_act_f32', lib.ptr(dy), co, lib.ptr(xbn), co, lib.ptr(dx), co, rows, co, sa, len(segs), lib.ptr(gamma), lib.ptr(mean_inv),
                     lib.ACT[act], 0.2, lib.ptr(sums), flag,
                     lib.ptr(dgamma), lib.ptr(dbeta), lib.ptr(dsum) if with_bias else None, 0, lib.ptr(dbias) if with_bias else None, st())
            torch.cuda.synchronize()
            out[mode] = [t.cpu().numpy().astype(np.float64) for t in (dy, dx, dgamma, dbeta, dbias)]
    finally:
        lib.call('tg_conv3x3_policy', was)
    scale = float(g_in.abs().max() * w.abs().max()) * 9 * ci
    names = ('dy', 'dx', 'dgamma', 'dbeta', 'dbias')
    for nm, a, b in zip(names, out['fused'], out['plain']):
        tol = 3e-5 * scale if nm == 'dy' else 3e-4 * max(1.0, np.abs(b).max())
        assert np.abs(a - b).max() <= tol, (nm, np.abs(a - b).max(), tol)
    assert not (out['fused'][1] == 7.0).any()


@pytest.mark.parametrize("train,use_mask", [(True, True), (True, False), (False, False)])
def test_mobn_apply_and_max_pool_in_one_launch(train, use_mask):
    """tg_mobn_apply_pool_f32 (the classifier's conv1_3 / conv2_3 -> max_pool -> dropout, Model/Good_GAN_cifar10.py:121-124,140-143) against
    tg_mobn_apply_f32 followed by tg_maxpool2_fwd_f32 on the same inputs: the same float operations in the same order — bit-identical activated
    tensor, pooled tensor and pop_mean, for a training pass over three application segments and for evaluation (sums = NULL, no mask)."""
    import ctypes as C
    lib = _lib()
    rng = np.random.default_rng(41)
    segs, h, w, c = [3, 2, 4], 8, 16, 128
    n = sum(segs)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    pop0 = rng.standard_normal(c).astype(np.float32)
    mask = (rng.random((n, h // 2, w // 2, c)) < 0.5).astype(np.float32)
    seg_rows = [s * h * w for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    sums = np.concatenate([x[o:o + s].astype(np.float64).sum(axis=(0, 1, 2)) for o, s in zip(np.cumsum([0] + segs[:-1]), segs)])
    sd = torch.from_numpy(sums).cuda() if train else None
    bd, md = dev(b), dev(mask) if use_mask else None
    xa, pa = dev(x), dev(pop0)
    oa = torch.full((n, h // 2, w // 2, c), 7.0, device='cuda')
    lib.call('tg_mobn_apply_f32', lib.ptr(xa), c, n * h * w, c, sa, len(segs), lib.ptr(sd), lib.ptr(bd), lib.ptr(pa), 0.9, lib.ACT['lrelu'], 0.2, st())
    lib.call('tg_maxpool2_fwd_f32', lib.ptr(xa), c, lib.ptr(oa), c, lib.ptr(md), c, 2.0, n, h, w, c, st())
    xb, pb = dev(x), dev(pop0)
    ob = torch.full((n, h // 2, w // 2, c), 7.0, device='cuda')
    lib.call('tg_mobn_apply_pool_f32', lib.ptr(xb), c, n, h, w, c, sa, len(segs), lib.ptr(sd), lib.ptr(bd), lib.ptr(pb), 0.9, lib.ACT['lrelu'], 0.2,
             lib.ptr(ob), c, lib.ptr(md), c, 2.0, st())
    np.testing.assert_array_equal(xb.cpu().numpy(), xa.cpu().numpy())
    np.testing.assert_array_equal(ob.cpu().numpy(), oa.cpu().numpy())
    np.testing.assert_array_equal(pb.cpu().numpy(), pa.cpu().numpy())
    if train:                                                  # and it is the oracle's arithmetic, not merely the same as the other kernel
        o, ref = 0, []
        for s_ in segs:
            ref.append(T.lrelu(x[o:o + s_].astype(np.float64) - x[o:o + s_].astype(np.float64).mean(axis=(0, 1, 2)) + b, 0.2))
            o += s_
        np.testing.assert_allclose(xb.cpu().numpy(), np.concatenate(ref), rtol=1e-5, atol=1e-5)
    bad = (C.c_int32 * 2)(5 * h * w - 7, 4 * h * w + 7)        # a segment boundary inside an image is refused
    with pytest.raises(lib.TgError, match='whole number'):
        lib.call('tg_mobn_apply_pool_f32', lib.ptr(xb), c, n, h, w, c, bad, 2, lib.ptr(sd), lib.ptr(bd), lib.ptr(pb), 0.9, lib.ACT['lrelu'], 0.2,
                 lib.ptr(ob), c, lib.ptr(md), c, 2.0, st())
