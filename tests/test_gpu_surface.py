"""The reference's Python surface on the HIP path (SURVEY §8b upper side, rows T3 / T4 / T21 / T22), against the float64 oracle:

  * `Model.forward_pass(...)` -> `Train_base._loss_GAN(D, C, Y, Lambda)` exactly as Training/Train_goodGAN.py:400-426,449-454 calls
    them — [G, D-list(6), C-list(5 | 4)] and the three losses, for Good_GAN_cifar10 and Good_GAN (MNIST, SVHN);
  * the helper heads of train_base.py:43-57,75-84,107 (value, gradient, accumulation) and `_loss_GAN` re-assembled from them;
  * activations called on a tensor (NN_Base._relu/_tanh/_leaky_relu/_softplus/_sigmoid, Good_GAN_cifar10.leakyReLu), forward + backward;
  * nn.batch_norm_impl, every flag combination of nn.conv2d_WN / dense_WN, the Salimans-style nn.dense / conv2d / deconv2d / nin
    (init=False and the data-dependent init=True branch) with variables created on first use (tf.get_variable semantics).
"""
import numpy as np
import pytest

from oracle import forward_pass as OF
from oracle import nets_goodgan as NG
from oracle import step_cifar10 as S
from oracle import step_goodgan as SG
from oracle import tf_ops as T
import gpu_common as G

pytestmark = pytest.mark.gpu
SIZES = dict(B_G=6, L_C=4, U_C=4, L_D=2, U_D=4)
LAMBDA = [0.3, 0.5]


def f64(d):
    return {k: (f64(v) if isinstance(v, dict) else np.asarray(v, np.float64)) for k, v in d.items()}


def _check_graph_outputs(got, ref, n_c):
    Gh, Dh, Ch = got
    Gr, Dr, Cr = ref
    assert G.rel_err(Gh.numpy().reshape(Gr.shape), Gr) < 2e-4
    assert len(Dh) == 6 and len(Ch) == n_c
    for i, (a, r) in enumerate(zip(Dh, Dr)):
        assert a.numpy().shape == r.shape, (i, a.numpy().shape, r.shape)
        assert np.abs(a.numpy() - r).max() <= 2e-4 * max(1.0, np.abs(r).max()), ('D', i)
    for i, (a, r) in enumerate(zip(Ch, Cr)):
        assert np.abs(a.numpy() - r).max() <= 2e-4 * max(1.0, np.abs(r).max()), ('C', i)
    for p, l in ((0, 1), (2, 3), (4, 5)):                       # the first element of each pair IS tf.nn.sigmoid of the second
        assert np.abs(Dh[p].numpy() - 1.0 / (1.0 + np.exp(-Dh[l].numpy().astype(np.float64)))).max() < 1e-6


def test_cifar10_forward_pass_then_loss_gan_as_the_reference_builds_its_graph():
    from tg.runtime import InjectedRNG
    P = S.init_params(3)
    full = dict(S.SIZES, **SIZES)
    b, r = S.synth_batch(11, full), S.synth_rnd(12, full)
    rnd = dict(C_real=r['C']['C_real'], C_unl=r['C']['C_unl'], C_unl_rep=r['C']['C_unl_rep'], C_unl_d=r['D']['C_unl_d'], C_fake=r['C']['C_fake'],
               D_real=r['D']['D_real'], D_fake=r['D']['D_fake'], D_unl=r['D']['D_unl'])
    zca = tuple(np.asarray(a, np.float64) for a in G.zca())
    ref, pops = OF.forward_pass_cifar10(f64(P), f64(b), f64(rnd), zca, True)
    ref_losses = OF.loss_gan(ref[1], ref[2], [b['y_g'].astype(np.float64), b['y_l_c'].astype(np.float64)], LAMBDA, True)

    tr = G.fresh_trainer(G.make_config(SIZES), P)
    cx = tr.cx
    inj = {}
    for k, v in G.cat_rnd(rnd['C_real'], rnd['C_unl'], rnd['C_unl_rep'], rnd['C_unl_d'], rnd['C_fake']).items():
        inj['fp/C/' + k] = v
    for k, v in G.cat_rnd(rnd['D_real'], rnd['D_fake'], rnd['D_unl']).items():
        inj['fp/D/' + k] = v
    cx.rng = InjectedRNG(inj, cx.device)
    tr.feed(b)
    PH = [tr.z_g_ph, tr.y_g_ph, tr.x_l_c_ph, tr.y_l_c_ph, tr.x_l_d_ph, tr.y_l_d_ph, tr.x_u_d_ph, tr.x_u_c_ph]
    with cx.phase_scope('fp', record=False):
        Gh, Dh, Ch = tr.model.forward_pass(*PH, True)                                            # Train_goodGAN.py:422
        d_loss, g_loss, c_loss = tr._goodGAN_loss(Gh, Dh, Ch, None, [tr.y_g_ph, tr.y_l_c_ph], LAMBDA, tr.model.discriminator)   # :72,449-454
    _check_graph_outputs((Gh, Dh, Ch), ref, 5)
    got = (float(d_loss), float(g_loss), float(c_loss))
    for a, e in zip(got, ref_losses):
        assert abs(a - e) <= 2e-4 * max(1.0, abs(e)), (got, ref_losses)
    st = cx.stores['classifier']
    for p, v in pops.items():                    # pop_mean after the five applications, call-site order (:228-240)
        assert G.rel_err(st.get(p + 'meanOnlyBatchNormalization/pop_mean'), v) < 2e-4, p
    # the same losses with Lambda as the trainer's device tensor
    tr.set_hyper(lambda_1=LAMBDA[0], lambda_2=LAMBDA[1])
    with cx.phase_scope('fp2', record=False):
        l2 = tr._loss_GAN(Dh, Ch, [tr.y_g_ph, tr.y_l_c_ph], tr.hyper[2:4])
    assert [float(v) for v in l2] == list(got)


@pytest.mark.parametrize("data", ['mnist', 'svhn'])
def test_goodgan_forward_pass_then_loss_gan(data):
    from Model.Good_GAN import Good_GAN
    from tg.runtime import InjectedRNG
    P = NG.init_params(data, 4)
    b, r = SG.synth_batch(data, 21, SIZES), SG.synth_rnd(data, 22, SIZES)
    rnd = dict(C_real=r['C']['C_real'], C_unl=r['C']['C_unl'], C_unl_d=r['D']['C_unl_d'], C_fake=r['C']['C_fake'],
               D_real=r['D']['D_real'], D_fake=r['D']['D_fake'], D_unl=r['D']['D_unl'])
    ref, _ = OF.forward_pass_goodgan(f64(P), data, f64(b), f64(rnd), True)
    ref_losses = OF.loss_gan(ref[1], ref[2], [b['y_g'].astype(np.float64), b['y_l_c'].astype(np.float64)], [0.1], False)

    tr = G.fresh_trainer(G.make_config_goodgan(data, SIZES), P, Good_GAN)
    cx = tr.cx
    inj = {}
    for k, v in G.cat_rnd(rnd['C_real'], rnd['C_unl'], rnd['C_unl_d'], rnd['C_fake']).items():
        inj['fp/C/' + k] = v
    for k, v in G.cat_rnd(rnd['D_real'], rnd['D_fake'], rnd['D_unl']).items():
        inj['fp/D/' + k] = v
    cx.rng = InjectedRNG(inj, cx.device)
    tr.feed(b)
    PH = [tr.z_g_ph, tr.y_g_ph, tr.x_l_c_ph, tr.y_l_c_ph, tr.x_l_d_ph, tr.y_l_d_ph, tr.x_u_d_ph, tr.x_u_c_ph]
    with cx.phase_scope('fp', record=False):
        Gh, Dh, Ch = tr.model.forward_pass(*PH, True)
        losses = tr._loss_GAN(Dh, Ch, [tr.y_g_ph, tr.y_l_c_ph], [0.1])
    _check_graph_outputs((Gh, Dh, Ch), ref, 4)
    got = tuple(float(v) for v in losses)
    for a, e in zip(got, ref_losses):
        assert abs(a - e) <= 5e-4 * max(1.0, abs(e)), (got, ref_losses)


# ------------------------------------------------------------------------------------------------ helper heads

@pytest.fixture(scope="module")
def plain():
    """a context + trainer object without caring about the networks (helper heads and free layer functions)."""
    tr = G.fresh_trainer(G.make_config(SIZES))
    return tr


def test_train_base_helper_heads_value_gradient_and_accumulation(plain):
    tr, cx = plain, plain.cx
    rng = np.random.default_rng(5)
    n = 37
    logits = (2.0 * rng.standard_normal((n, 10))).astype(np.float32)
    labels = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    z = rng.standard_normal((n, 1)).astype(np.float32)
    zt = (rng.random((n, 1)) < 0.5).astype(np.float32)
    l64, z64 = logits.astype(np.float64), z.astype(np.float64)
    with cx.phase_scope('heads', record=False):
        cases = [
            (lambda a: tr._entropy(a), logits, T.entropy(l64)),
            (lambda a: tr._balance_entropy(a), logits, T.balance_entropy(l64)),
            (lambda a: tr._softmax_cross_entropy_loss_w_logits(cx.from_numpy(labels), a), logits, T.softmax_ce_mean(l64, labels.astype(np.float64))),
            (lambda a: tr._sigmoid_cross_entopy_w_logits(1.0, a), z, T.bce_mean(z64, np.ones_like(z64))),
            (lambda a: tr._sigmoid_cross_entopy_w_logits(0.0, a), z, T.bce_mean(z64, np.zeros_like(z64))),
            (lambda a: tr._sigmoid_cross_entopy_w_logits(cx.from_numpy(zt), a), z, T.bce_mean(z64, zt.astype(np.float64))),
        ]
        for i, (fn, x, (v_ref, g_ref)) in enumerate(cases):
            a = cx.from_numpy(x, ld=32 if x.shape[1] == 10 else None)
            v = fn(a)
            assert abs(float(v) - v_ref) <= 2e-6 * max(1.0, abs(v_ref)), (i, float(v), v_ref)
            assert np.abs(a.grad.numpy() - g_ref).max() <= 2e-6 * max(np.abs(g_ref).max(), 1e-3), i
            v2 = fn(a)                                           # a second head on the same tensor ADDS its gradient
            assert float(v2) == float(v)
            assert np.abs(a.grad.numpy() - 2 * g_ref).max() <= 4e-6 * max(np.abs(g_ref).max(), 1e-3), i
        # weight scales value and gradient
        a = cx.from_numpy(logits, ld=32)
        v = tr._entropy(a, weight=0.25)
        assert abs(float(v) - 0.25 * T.entropy(l64)[0]) < 1e-6 and np.abs(a.grad.numpy() - 0.25 * T.entropy(l64)[1]).max() < 1e-6


def test_loss_gan_equals_its_reassembly_from_the_helper_heads(plain):
    """train_base.py:113-154 written with the helper methods, as the reference writes it, equals the fused heads."""
    tr, cx = plain, plain.cx
    rng = np.random.default_rng(6)
    mk = lambda n, c: rng.standard_normal((n, c)).astype(np.float32)
    d_real, d_fake, d_unl = mk(6, 1), mk(6, 1), mk(4, 1)
    c_real, c_unl, c_unl_d, c_fake, c_rep = mk(4, 10), mk(4, 10), mk(4, 10), mk(6, 10), mk(4, 10)
    y_g = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 6)]
    y_l = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 4)]
    ref = OF.loss_gan([None, d_real.astype(np.float64), None, d_fake.astype(np.float64), None, d_unl.astype(np.float64)],
                      [a.astype(np.float64) for a in (c_real, c_unl, c_unl_d, c_fake, c_rep)], [y_g.astype(np.float64), y_l.astype(np.float64)], LAMBDA, True)
    with cx.phase_scope('lg', record=False):
        A = lambda x: cx.from_numpy(x, ld=32 if x.shape[1] == 10 else None)     # logits: channel-padded as the networks emit them
        Y = lambda y: cx.from_numpy(y)                                          # labels: dense [n][10] (the placeholders' layout)
        D = [None, A(d_real), None, A(d_fake), None, A(d_unl)]
        C = [A(c_real), A(c_unl), A(c_unl_d), A(c_fake), A(c_rep)]
        fused = [float(v) for v in tr._loss_GAN(D, C, [Y(y_g), Y(y_l)], LAMBDA)]
        dr, df, du = A(d_real), A(d_fake), A(d_unl)
        d_loss = float(tr._sigmoid_cross_entopy_w_logits(1.0, dr)) + 0.5 * float(tr._sigmoid_cross_entopy_w_logits(0.0, df)) \
            + 0.5 * float(tr._sigmoid_cross_entopy_w_logits(0.0, du))
        g_loss = 0.5 * float(tr._sigmoid_cross_entopy_w_logits(1.0, A(d_fake)))
        cu = A(c_unl)
        c_real_term = float(tr._softmax_cross_entropy_loss_w_logits(Y(y_l), A(c_real))) + 1e-6 * float(tr._entropy(cu)) + 1e-3 * float(tr._balance_entropy(cu))
        c_fake_term = float(tr._softmax_cross_entropy_loss_w_logits(Y(y_g), A(c_fake)))
    for a, e in zip(fused, ref):
        assert abs(a - e) <= 2e-5 * max(1.0, abs(e)), (fused, ref)
    assert abs(d_loss - ref[0]) < 2e-5 and abs(g_loss - ref[1]) < 2e-5
    c_unl_term = T.c_unl_loss(c_unl.astype(np.float64), d_unl.astype(np.float64))[0]
    mse = T.mse_mean(c_unl.astype(np.float64), c_rep.astype(np.float64))[0]
    assert abs(0.005 * c_unl_term + c_real_term + LAMBDA[0] * c_fake_term + LAMBDA[1] * mse - ref[2]) < 5e-5


def test_metric_streams_accuracy(plain):
    tr, cx = plain, plain.cx
    rng = np.random.default_rng(7)
    logits = rng.standard_normal((50, 10)).astype(np.float32)
    labels = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 50)]
    labels[:20] = np.eye(10, dtype=np.float32)[logits[:20].argmax(1)]
    want = float((logits.argmax(1) == labels.argmax(1)).mean())
    with cx.phase_scope('met', record=False):
        acc, update_op, reset_op, pred, probs = tr._metric(cx.from_numpy(logits[:30], ld=32), cx.from_numpy(labels[:30]))
        update_op(cx.from_numpy(labels[30:]), cx.from_numpy(logits[30:], ld=32))
        assert abs(float(acc) - want) < 1e-7
        np.testing.assert_array_equal(pred.cpu().numpy().reshape(30, 10), np.eye(10, dtype=np.float32)[logits[:30].argmax(1)])
        reset_op()
        assert float(acc) == 0.0


# ------------------------------------------------------------------------------------------------ activations as ops

_NP_ACT = {'_relu': lambda x: np.maximum(x, 0), '_tanh': np.tanh, '_leaky_relu': lambda x: np.where(x > 0, x, 0.2 * x),
           '_softplus': lambda x: np.log1p(np.exp(x)), '_sigmoid': lambda x: 1 / (1 + np.exp(-x)), 'leakyReLu': lambda x: np.maximum(x, 0) - 0.2 * np.maximum(-x, 0)}
_NP_DACT = {'_relu': lambda x: (x > 0) * 1.0, '_tanh': lambda x: 1 - np.tanh(x) ** 2, '_leaky_relu': lambda x: np.where(x > 0, 1.0, 0.2),
            '_softplus': lambda x: 1 / (1 + np.exp(-x)), '_sigmoid': lambda x: np.exp(-x) / (1 + np.exp(-x)) ** 2, 'leakyReLu': lambda x: np.where(x > 0, 1.0, 0.2)}


@pytest.mark.parametrize("name", sorted(_NP_ACT))
def test_activations_called_on_a_tensor(plain, name):
    tr, cx = plain, plain.cx
    rng = np.random.default_rng(8)
    for shape, ld in (((3, 5, 5, 7), None), ((4, 2, 2, 32), None), ((9, 1), None), ((5, 10), 32)):
        x = rng.standard_normal(shape).astype(np.float32)
        dy = rng.standard_normal(shape).astype(np.float32)
        with cx.phase_scope('act', train_nets=('x',)):
            a = cx.from_numpy(x, ld=ld)
            a.requires_grad = True
            y = getattr(tr.model, name)(a)
            y.grad = cx.from_numpy(dy, ld=ld)
            cx.backward()
        x64 = x.astype(np.float64)
        assert np.abs(y.numpy().reshape(shape) - _NP_ACT[name](x64)).max() < 2e-6
        assert np.abs(a.grad.numpy().reshape(shape) - dy * _NP_DACT[name](x64)).max() < 5e-6
    # a non-default slope when called (the fused form always uses 0.2)
    if name in ('_leaky_relu', 'leakyReLu'):
        x = rng.standard_normal((4, 8)).astype(np.float32)
        with cx.phase_scope('act2', record=False):
            y = getattr(tr.model, name)(cx.from_numpy(x), 0.05)
        assert np.abs(y.numpy() - np.where(x > 0, x, 0.05 * x)).max() < 1e-6


# ------------------------------------------------------------------------------------------------ free layer functions

def _bwd(cx, root, fn, x, dy_fn):
    """run fn on Act(x) with gradients: returns (y array, dx array, {leaf: grad})."""
    with cx.phase_scope('L_' + root, train_nets=(root,)):
        with cx.variable_scope(root):
            a = cx.from_numpy(x, ld=(x.shape[-1] + 31) // 32 * 32)
            a.requires_grad = True
            y = fn(a)
            dy = dy_fn(y.numpy().shape)
            y.grad = cx.from_numpy(dy, ld=y.ld)
            cx.backward()
    st = cx.stores[root]
    return y.numpy(), a.grad.numpy(), {k: st.get(k, 'grad') for k in st.names(True)}, dy


def test_batch_norm_impl_train_and_eval(plain):
    from Model import nn
    cx = plain.cx
    rng = np.random.default_rng(9)
    x = (1.5 * rng.standard_normal((6, 4, 4, 32)) + 0.7).astype(np.float32)
    y, dx, grads, dy = _bwd(cx, 'bnroot', lambda a: nn.batch_norm_impl(a, is_conv_out=True, deterministic=False), x, lambda s: rng.standard_normal(s).astype(np.float32))
    x64 = x.astype(np.float64)
    yr, cache = T.batch_norm_train(x64, np.ones(32), np.zeros(32), 0.001)
    assert G.rel_err(y, yr) < 2e-5
    dxr, dgr, dbr = T.batch_norm_train_bwd(dy.astype(np.float64), np.ones(32), cache)
    assert G.rel_err(dx, dxr) < 1e-4
    assert G.rel_err(grads['bnroot/BatchNormalization/scale'], dgr) < 1e-4 and G.rel_err(grads['bnroot/BatchNormalization/beta'], dbr) < 1e-4
    st = cx.stores['bnroot']
    mu, var = x64.mean((0, 1, 2)), x64.var((0, 1, 2))                      # tf.nn.moments: biased variance (nn.py:205-213)
    assert G.rel_err(st.get('bnroot/BatchNormalization/pop_mean'), 0.1 * mu) < 1e-5
    assert G.rel_err(st.get('bnroot/BatchNormalization/pop_var'), 0.9 + 0.1 * var) < 1e-5
    with cx.phase_scope('bn_eval', record=False):
        with cx.variable_scope('bnroot'):
            ye = nn.batch_norm_impl(cx.from_numpy(x), deterministic=True).numpy()
    assert G.rel_err(ye, (x64 - 0.1 * mu) / np.sqrt(0.9 + 0.1 * var + 0.001)) < 2e-5


def _conv_ref(x64, V, g, b, k, stride, pad):
    W = T.wn_weight(V, g) if g is not None else V
    y = T.conv2d(x64, W.reshape(k, k, x64.shape[-1], -1), (stride, stride), pad)
    return y + (b if b is not None else 0.0)


@pytest.mark.parametrize("flags", ['wn', 'plain', 'bn', 'wn_mobn'])
def test_conv2d_wn_flag_combinations(plain, flags):
    """nn.conv2d_WN (nn.py:469-520): weight norm + bias, plain conv + bias, conv + batch_norm_impl, weight norm + mean-only BN — with a
    relu fused or applied behind, variables created on first use with the reference's initialisers, gradients of every variable."""
    from Model import nn
    tr, cx = plain, plain.cx
    rng = np.random.default_rng(10)
    x = rng.standard_normal((4, 6, 6, 5)).astype(np.float32)
    kw = dict(wn=dict(use_weight_normalization=True), plain={}, bn=dict(use_batch_normalization=True),
              wn_mobn=dict(use_weight_normalization=True, use_mean_only_batch_normalization=True))[flags]
    root = 'cw_' + flags
    fn = lambda a: nn.conv2d_WN(a, 32, filter_size=[3, 3], stride=[2, 2] if flags == 'wn' else [1, 1], nonlinearity=tr.model._relu, name='L', **kw)
    y, dx, grads, dy = _bwd(cx, root, fn, x, lambda s: rng.standard_normal(s).astype(np.float32))
    st = cx.stores[root]
    V = st.get(root + '/L/V').astype(np.float64)
    assert V.shape == (3, 3, 5, 32) and abs(V.std() - 0.05) < 0.01                           # random_normal_initializer(0, 0.05), nn.py:478
    x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
    stride = 2 if flags == 'wn' else 1
    g = np.ones(32) if 'wn' in flags else None
    pre = _conv_ref(x64, V, g, None, 3, stride, 'SAME')
    if flags == 'bn':
        z, cache = T.batch_norm_train(pre, np.ones(32), np.zeros(32), 0.001)
    elif flags == 'wn_mobn':
        z = pre - pre.mean((0, 1, 2))
    else:
        z = pre
    yr = np.maximum(z, 0)
    assert G.rel_err(y, yr) < 1e-4
    dz = dy64 * (yr > 0)
    if flags == 'bn':
        dpre, dgam, dbeta = T.batch_norm_train_bwd(dz, np.ones(32), cache)
        assert G.rel_err(grads[root + '/L/BatchNormalization/scale'], dgam) < 1e-3
        assert root + '/L/b' not in st.index                                               # no bias with batch norm (nn.py:480-482)
    elif flags == 'wn_mobn':
        dpre = dz - dz.mean((0, 1, 2))
        assert G.rel_err(grads[root + '/L/b'], dz.sum((0, 1, 2))) < 1e-3
    else:
        dpre = dz
        assert G.rel_err(grads[root + '/L/b'], dz.sum((0, 1, 2))) < 1e-3
    W = T.wn_weight(V, g) if g is not None else V
    dW = T.conv2d_bwd_filter(x64, dpre, W.shape, (stride, stride), 'SAME')
    dxr = T.conv2d_bwd_input(x64.shape, W, dpre, (stride, stride), 'SAME')
    assert G.rel_err(dx, dxr) < 1e-3
    if g is not None:
        dV, dg = T.wn_weight_bwd(V, g, dW)
        assert G.rel_err(grads[root + '/L/V'], dV) < 1e-3 and G.rel_err(grads[root + '/L/g'], dg) < 1e-3
    else:
        assert G.rel_err(grads[root + '/L/V'], dW) < 1e-3


def test_salimans_layers_and_data_dependent_init(plain):
    """nn.dense / conv2d / deconv2d / nin (nn.py:220-340): counters name the layers, init=False is g*op(x, l2_normalize(V)) + b,
    init=True returns scale_init*(x_init - m_init)."""
    from Model import nn
    tr, cx = plain, plain.cx
    rng = np.random.default_rng(11)
    counters = {}
    x4 = rng.standard_normal((3, 4, 4, 6)).astype(np.float32)
    x2 = rng.standard_normal((5, 12)).astype(np.float32)
    with cx.phase_scope('sal', record=False):
        with cx.variable_scope('sal'):
            A4, A2 = cx.from_numpy(x4, ld=32), cx.from_numpy(x2, ld=32)
            yc = nn.conv2d(A4, 32, nonlinearity=tr.model._leaky_relu, counters=counters)
            yc2 = nn.conv2d(A4, 32, filter_size=[3, 3], stride=[2, 2], counters=counters)
            yd = nn.dense(A2, 32, counters=counters)
            yn = nn.nin(A4, 32, nonlinearity=tr.model._relu, counters=counters)
            yt = nn.deconv2d(A4, 32, filter_size=[5, 5], stride=[2, 2], counters=counters)
            yi = nn.conv2d(A4, 32, init=True, init_scale=0.7, counters={'conv2d': 0})                    # reuses conv2d_0's V
            ydi = nn.dense(A2, 32, init=True, counters={'dense': 0})
            with pytest.raises(ValueError, match="only filter_size"):
                nn.deconv2d(A4, 32, counters={})
    assert counters == {'conv2d': 2, 'dense': 2, 'deconv2d': 1}
    st = cx.stores['sal']
    for leaf in ('conv2d_0', 'conv2d_1', 'dense_0', 'dense_1', 'deconv2d_0'):
        assert {'sal/%s/V' % leaf, 'sal/%s/g' % leaf, 'sal/%s/b' % leaf} <= set(st.index), leaf
    P = lambda n: st.get(n).astype(np.float64)
    x4d, x2d = x4.astype(np.float64), x2.astype(np.float64)
    ones = np.ones(32)
    r = _conv_ref(x4d, P('sal/conv2d_0/V'), ones, 0.0, 3, 1, 'SAME')
    assert G.rel_err(yc.numpy(), np.where(r > 0, r, 0.2 * r)) < 1e-4
    assert G.rel_err(yc2.numpy(), _conv_ref(x4d, P('sal/conv2d_1/V'), ones, 0.0, 3, 2, 'SAME')) < 1e-4
    assert G.rel_err(yd.numpy(), x2d @ T.wn_weight(P('sal/dense_0/V'), ones)) < 1e-4
    rn = (x4d.reshape(-1, 6) @ T.wn_weight(P('sal/dense_1/V'), ones)).reshape(3, 4, 4, 32)
    assert G.rel_err(yn.numpy(), np.maximum(rn, 0)) < 1e-4
    Vt = P('sal/deconv2d_0/V')
    assert Vt.shape == (5, 5, 32, 6)
    assert G.rel_err(yt.numpy(), T.conv2d_transpose(x4d, T.wn_weight(Vt, ones, 2))) < 1e-4
    xi = _conv_ref(x4d, P('sal/conv2d_0/V'), ones, None, 3, 1, 'SAME')
    assert G.rel_err(yi.numpy(), 0.7 / np.sqrt(xi.var((0, 1, 2)) + 1e-8) * (xi - xi.mean((0, 1, 2)))) < 1e-4
    xd = x2d @ T.wn_weight(P('sal/dense_0/V'), ones)
    assert G.rel_err(ydi.numpy(), 1.0 / np.sqrt(xd.var(0) + 1e-10) * (xd - xd.mean(0))) < 1e-4


def test_fused_backward_paths_refuse_a_second_consumer(plain):
    """advisor (round 3): the batch-norm backward fusions assume ONE consumer per tensor.  (a) conv -> relu -> x, consumed by a batch norm AND by
    a second convolution: the batch norm's backward folds relu'(x) and the bias gradient into x's gradient buffer (Act.grad_is_dpre), the other
    consumer writes a raw gradient into the same buffer — the producing convolution must refuse to train on that buffer (Act.contribs, counted
    in Context.grad_of) instead of treating it as its pre-activation gradient.  (b) a batch norm's OUTPUT consumed by two convolutions: the
    backward statistics the first consumer's input-gradient launch took (Act.bn_bwd_sums) are not the statistics of the final gradient — the
    batch norm must fall back to its own statistics pass, and then gives the oracle's gradient for the gradient buffer as it stands."""
    from tg import lib, ops
    cx = plain.cx
    rng = np.random.default_rng(21)
    n, h, c = 4, 8, 32
    xin = rng.standard_normal((n, h, h, c)).astype(np.float32)
    mk = lambda *s: cx.from_numpy((0.1 * rng.standard_normal(s)).astype(np.float32).reshape(1, -1)).t
    zeros = lambda *s: cx.from_numpy(np.zeros(s, np.float32).reshape(1, -1)).t
    wa, ba, wb, wc = mk(3, 3, c, c), mk(c), mk(3, 3, c, c), mk(3, 3, c, c)
    gwa, gba, gwb, gwc = zeros(3, 3, c, c), zeros(c), zeros(3, 3, c, c), zeros(3, 3, c, c)
    gamma, beta, mm, mv = cx.from_numpy(np.ones((1, c), np.float32)).t, zeros(c), zeros(c), cx.from_numpy(np.ones((1, c), np.float32)).t
    ggam, gbet = zeros(c), zeros(c)

    def graph(second_consumer_of):
        a = cx.from_numpy(xin)
        a.requires_grad = True
        x = ops.conv2d(a, wa, ba, c, 3, 1, 'SAME', act='relu', kernel_grad=gwa, bias_grad=gba)
        y = ops.batch_norm_train(x, gamma, beta, mm, mv, 1e-3, 0.9, gamma_grad=ggam, beta_grad=gbet)
        u = ops.conv2d(y, wb, None, c, 3, 1, 'SAME', kernel_grad=gwb)
        v = ops.conv2d(x if second_consumer_of == 'x' else y, wc, None, c, 3, 1, 'SAME', kernel_grad=gwc)
        du = rng.standard_normal((n, h, h, c)).astype(np.float32)
        u.grad, v.grad = cx.from_numpy(du), cx.from_numpy(du)
        return x, y

    with cx.phase_scope('twocons_a', train_nets=('discriminator',)):      # any store name: only cx.trains() matters
        with cx.variable_scope('discriminator'):
            graph('x')
            with pytest.raises(lib.TgError, match='written by 2 consumers'):
                cx.backward()
    with cx.phase_scope('twocons_b', train_nets=('discriminator',)):
        with cx.variable_scope('discriminator'):
            x, y = graph('y')
            cx.backward()
            assert y.grad.contribs == 2 and y.bn_bwd_sums is None          # the second consumer withdrew the first one's statistics
            gy = y.grad.numpy().astype(np.float64)
            xr = x.numpy().astype(np.float64)
            _, cache = T.batch_norm_train(xr, np.ones(c), np.zeros(c), 1e-3)
            dxr, dgr, dbr = T.batch_norm_train_bwd(gy, np.ones(c), cache)
            assert G.rel_err(ggam.cpu().numpy(), dgr) < 1e-4 and G.rel_err(gbet.cpu().numpy(), dbr) < 1e-4
