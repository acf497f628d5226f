"""GPU parity of SURVEY §8f N4: the loss variants of Training/train_base.py:156-574 (host composition in the package's
Training/train_base.py over tg_c_loss_terms_f32 / tg_true_fake_loss_f32 / tg_sqdiff_rows_loss_f32 / tg_d_loss_terms_f32 and the
feature-matching / pull-away kernels) and minibatch discrimination (Model/modle_base.py:110-128), against the float64 restatement
oracle/loss_variants.py on identical float32 inputs: every returned value, every gradient."""
import numpy as np
import pytest

from oracle import loss_variants as L
import gpu_common as G
from test_oracle_loss_variants import LAMBDA, VARIANTS, call, make_inputs

pytestmark = pytest.mark.gpu
SIZES = dict(B_G=6, L_C=4, U_C=4, L_D=2, U_D=4)


def flat(v):
    out = []
    for e in (v if isinstance(v, (list, tuple)) else [v]):
        out += flat(e) if isinstance(e, (list, tuple)) else [float(e)]
    return out


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("variant", VARIANTS)
def test_loss_variant_values_and_gradients(variant, fast):
    if fast and variant != 'GoodRegGAN_BS_cifar10':
        pytest.skip("FAST_MODE only changes _loss_GoodRegGAN_BS_cifar10")
    from Training.train_base import Train_base
    tr = G.fresh_trainer(G.make_config(SIZES))
    cx = tr.cx
    x32 = make_inputs(11, np.float32)
    x64 = {k: v.astype(np.float64) for k, v in x32.items()}
    kw = dict(fast_mode=True) if fast else {}
    # the oracle in its own nesting (call() flattens d_loss / c_loss to their totals: take the raw return here)
    D64 = [None, x64['d_real'], None, x64['d_fake'], None, x64['d_unl']]
    _, grads, m = call(variant, x64, **kw)
    tb = Train_base()
    tb.config = type('Cfg', (), {'FAST_MODE': fast})()
    with cx.phase_scope('T', record=False):
        a = {k: cx.from_numpy(v, ld=(32 if v.shape[1] in (1, 10) else None)) for k, v in x32.items() if k not in ('y_l_c', 'y_g')}
        a['y_l_c'], a['y_g'] = cx.from_numpy(x32['y_l_c']), cx.from_numpy(x32['y_g'])
        D = [None, a['d_real'], None, a['d_fake'], None, a['d_unl']]
        Y = [a['y_g'], a['y_l_c']]
        if variant == 'BGAN':
            got = tb._loss_BGAN([a['c_real'], a['c_unl'], a['c_fake'], a['f_real'], a['f_unl'], a['f_bfake']], [a['y_l_c']])
            ref = L.loss_BGAN([x64['c_real'], x64['c_unl'], x64['c_fake'], x64['f_real'], x64['f_unl'], x64['f_bfake']], [x64['y_l_c']])[0]
        else:
            base = [a['c_real'], a['c_unl'], a['c_unl_d'], a['c_gfake'], a['c_bfake']]
            base64 = [x64['c_real'], x64['c_unl'], x64['c_unl_d'], x64['c_gfake'], x64['c_bfake']]
            Y64 = [x64['y_g'], x64['y_l_c']]
            fe = lambda d, *ks: [d[k] for k in ks]
            if variant == 'GoodBadGAN':
                ks = ('f_real', 'f_unl', 'f_bfake')
                got = tb._loss_GoodBadGAN(D, base + fe(a, *ks), Y, LAMBDA[:1])
                ref = L.loss_GoodBadGAN(D64, base64 + fe(x64, *ks), Y64, LAMBDA[:1])[0]
            elif variant == 'GoodRegBadGAN':
                ks = ('c_pert', 'f_real', 'f_unl', 'f_bfake', 'f_pert')
                got = tb._loss_GoodRegBadGAN(D, base + fe(a, *ks), Y, LAMBDA[:1])
                ref = L.loss_GoodRegBadGAN(D64, base64 + fe(x64, *ks), Y64, LAMBDA[:1])[0]
            elif variant == 'GoodRegGAN':
                ks = ('c_pert', 'f_real', 'f_unl', 'f_bfake', 'f_pert')
                got = tb._loss_GoodRegGAN(D, base + fe(a, *ks), Y, LAMBDA[:3])
                ref = L.loss_GoodRegGAN(D64, base64 + fe(x64, *ks), Y64, LAMBDA[:3])[0]
            elif variant == 'GoodRegGAN_cifar10':
                ks = ('c_pert', 'f_real', 'f_unl', 'f_bfake', 'f_pert', 'c_rep')
                got = tb._loss_GoodRegGAN_cifar10(D, base + fe(a, *ks), Y, LAMBDA)
                ref = L.loss_GoodRegGAN_cifar10(D64, base64 + fe(x64, *ks), Y64, LAMBDA)[0]
            elif variant == 'GoodRegGAN_BS':
                ks = ('c_pert', 'c_unl_bg', 'f_real', 'f_unl', 'f_bfake', 'f_pert', 'f_unl_bg')
                got = tb._loss_GoodRegGAN_BS(D, base + fe(a, *ks), Y, LAMBDA[:3])
                ref = L.loss_GoodRegGAN_BS(D64, base64 + fe(x64, *ks), Y64, LAMBDA[:3])[0]
            else:
                ks = ('c_pert', 'c_unl_bg', 'f_real', 'f_unl', 'f_bfake', 'f_pert', 'f_unl_bg', 'c_rep')
                got = tb._loss_GoodRegGAN_BS_cifar10(D, base + fe(a, *ks), Y, LAMBDA)
                ref = L.loss_GoodRegGAN_BS_cifar10(D64, base64 + fe(x64, *ks), Y64, LAMBDA, **kw)[0]
        gv, rv = flat(got), flat(ref)
        assert len(gv) == len(rv), (len(gv), len(rv))                      # same nesting / number of reported terms as the reference
        np.testing.assert_allclose(gv, rv, rtol=2e-5, atol=2e-6)
        # gradients: what each solver differentiates
        dkeys = {'d_real': 0, 'd_fake': 1, 'd_unl': 2}
        for gk, (xk, _) in m.items():
            ref_g = grads[gk]
            if gk in dkeys:                                              # d_loss's gradient lives on the concatenated logits
                n0 = [0, x32['d_real'].shape[0], x32['d_real'].shape[0] + x32['d_fake'].shape[0]][dkeys[gk]]
                got_g = tb.last_d_cat.grad.numpy()[n0:n0 + ref_g.shape[0]]
            elif gk == 'c_gfake' and fast:
                assert a['c_gfake'].grad is None                         # the term is dropped: no gradient reaches those logits
                continue
            elif gk == 'c_rep' and variant == 'GoodRegGAN_BS_cifar10':
                assert a['c_rep'].grad is None                           # train_base.py:503: the "unsupervised loss" is the constant lambda_4
                continue
            else:
                got_g = a[{'gG_d_fake': 'd_fake', 'feat_bfake': 'f_bfake', 'feat_fake': 'f_bfake'}.get(gk, xk)].grad.numpy()
            scale = max(np.abs(ref_g).max(), 1e-12)
            assert np.abs(got_g - ref_g).max() <= 2e-5 * scale + 1e-9, (variant, gk, np.abs(got_g - ref_g).max(), scale)


def test_minibatch_discrimination_forward_backward():
    from tg import ops
    tr = G.fresh_trainer(G.make_config(SIZES))
    cx = tr.cx
    r = np.random.default_rng(3)
    n, f, k, d = 37, 138, 100, 5
    x = r.standard_normal((n, f)).astype(np.float32)
    w = (0.1 * r.standard_normal((f, k * d))).astype(np.float32)
    b = r.standard_normal(k).astype(np.float32)
    dout = r.standard_normal((n, f + k)).astype(np.float32)
    ref, cache = L.minibatch_discrimination(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), d)
    dx, dw, db = L.minibatch_discrimination_bwd(x.astype(np.float64), w.astype(np.float64), cache, dout[:, f:].astype(np.float64))
    dx = dx + dout[:, :f]                                                # the concatenated input passes its gradient through
    import torch
    wt, bt = torch.from_numpy(w).cuda().reshape(-1), torch.from_numpy(b).cuda()
    gw, gb = torch.zeros_like(wt), torch.zeros_like(bt)
    with cx.phase_scope('T', train_nets=('discriminator',)), cx.variable_scope('discriminator'):
        xa = cx.from_numpy(x, ld=160)
        xa.requires_grad = True
        out = ops.minibatch_discrimination(xa, wt, bt, k, d, w_grad=gw, b_grad=gb, concat_input=True)
        got = out.numpy()
        out.grad = cx.from_numpy(dout, ld=out.ld)
        cx.backward()
    assert out.c == f + k and np.array_equal(got[:, :f], x)
    assert G.rel_err(got[:, f:], ref) < 2e-5
    assert G.rel_err(xa.grad.numpy(), dx) < 5e-5
    assert G.rel_err(gw.cpu().numpy().reshape(f, k * d), dw) < 5e-5
    assert G.rel_err(gb.cpu().numpy(), db) < 2e-5
    # f alone (the reference's return value)
    with cx.phase_scope('T2', record=False):
        f_only = ops.minibatch_discrimination(cx.from_numpy(x, ld=160), wt, bt, k, d)
        assert f_only.c == k and G.rel_err(f_only.numpy(), ref) < 2e-5


def test_svhn_discriminator_with_minibatch_discrimination():
    """Model/Good_GAN.py:159-162 (MINIBATCH_DIS, off in every config of the reference): variables discriminator/w, discriminator/b,
    discriminator/d_h3_lin/d_h3_lin/{kernel,bias}; the logits' directional derivative along the gradient of each of those variables and
    of the last convolution's filter agrees with a central difference of the forward pass."""
    import torch
    from Model.Good_GAN import Good_GAN
    tr = G.fresh_trainer(G.make_config_goodgan('svhn', SIZES, MINIBATCH_DIS=True), None, Good_GAN)
    cx, m, st = tr.cx, tr.model, tr.cx.stores['discriminator']
    names = st.names(True)
    assert 'discriminator/w' in names and 'discriminator/b' in names and 'discriminator/d_h3_lin/d_h3_lin/kernel' in names
    assert 'discriminator/d_h3_wndense/V' not in names and st.get('discriminator/w').shape == (138, 500)
    rng = np.random.default_rng(0)
    for k in names:                                                      # benign weights (the reference's initialisers saturate: SURVEY T18)
        if k.endswith(('/V', 'kernel', '/w')):
            st.set(k, 0.05 * rng.standard_normal(st.get(k).shape))
    n = 6
    img = rng.uniform(-1, 1, (n, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    r = rng.standard_normal((n, 1)).astype(np.float32)
    from tg.runtime import InjectedRNG
    masks = {'T/D/drop0': np.ones((n, 32, 32, 3), np.float32), 'T/D/drop1': np.ones((n, 16, 16, 32), np.float32),
             'T/D/drop2': np.ones((n, 8, 8, 64), np.float32)}

    def forward(train):
        cx.rng = InjectedRNG(dict(masks), cx.device)
        scope = cx.phase_scope('T', train_nets=('discriminator',)) if train else cx.phase_scope('T', record=False)
        with scope:
            with cx.rng_scoped('T/D'):
                _, lg = m.discriminator(cx.from_numpy(img), cx.from_numpy(y))
            if train:
                lg.grad = cx.from_numpy(r, ld=32)
                cx.backward()
            return float((lg.numpy() * r).sum())

    st.g.zero_()
    forward(True)
    torch.cuda.synchronize()
    for name in ('discriminator/w', 'discriminator/b', 'discriminator/d_h3_lin/d_h3_lin/kernel', 'discriminator/d_h2_wnconv1/V'):
        p0 = st.get(name).copy()
        g = st.get(name, 'grad')
        direction = (g / np.linalg.norm(g)).astype(np.float32)           # steepest ascent: the largest signal over the fp32 noise of f
        h = 1e-2
        st.set(name, p0 + h * direction)
        fp = forward(False)
        st.set(name, p0 - h * direction)
        fm = forward(False)
        st.set(name, p0)
        fd, an = (fp - fm) / (2 * h), float((g * direction).sum())
        assert abs(fd - an) <= 3e-2 * max(abs(fd), abs(an), 1e-3), (name, fd, an)
