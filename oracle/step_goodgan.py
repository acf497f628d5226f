"""ORACLE — test infrastructure only (see oracle/tf_ops.py header; PARITY UNPINNED).

One Triple-GAN iteration (D-update, G-update, C-update + EMA) of Training/Train_goodGAN.py:266-276 for the
MNIST / SVHN models of Model/Good_GAN.py (forward_pass :428-472: no ZCA, four classifier outputs, no
consistency term).  Same structure as oracle/step_cifar10.py; batch-norm moving statistics of the classifier
are updated sequentially in call-site order (C_real, C_unl, C_unl_d, C_fake) restricted to what a solver runs.
"""
import numpy as np
from . import tf_ops as T
from . import nets_goodgan as N
from .step_cifar10 import _adam


def new_state(P):
    st = {'P': {k: v.copy() for k, v in P.items()}, 't': {'D': 0, 'G': 0, 'C': 0}}
    tr = [k for k in P if 'moving_' not in k]
    st['m'] = {k: np.zeros_like(P[k]) for k in tr}
    st['v'] = {k: np.zeros_like(P[k]) for k in tr}
    st['ema'] = {k: P[k].copy() for k in tr if k.startswith('classifier/')}
    return st


def _gen(P, data, b, bn_updates=None):
    return N.seq_fwd(P, N.generator_layers(data), b['z_g'], b['y_g'], {}, True, bn_updates)


def d_phase(st, data, b, rnd, hyper, labels=None):
    """labels (tests only): {'unl': one-hot [U_C,10], 'unl_d': one-hot [U_D,10]} to use INSTEAD of the arg-max of the classifier's logits —
    lets a parity test give the oracle's discriminator the labels the implementation under test fed its own, when a near-tie arg-max
    (random-init logits, bf16 operand noise) came out differently on the two sides."""
    P = st['P']
    CL, DL = N.classifier_layers(data), N.discriminator_layers(data)
    bnu = {}
    Gimg, _, _ = _gen(P, data, b, bnu)
    c_unl, _, _ = N.seq_fwd(P, CL, b['x_u_c'], None, rnd['C_unl'], True, bnu)
    c_unl_d, _, _ = N.seq_fwd(P, CL, b['x_u_d'], None, rnd['C_unl_d'], True, bnu)
    N.commit_bn(P, bnu)
    oh_unl = T.argmax_onehot(c_unl) if labels is None else np.asarray(labels['unl'], c_unl.dtype)
    oh_unl_d = T.argmax_onehot(c_unl_d) if labels is None else np.asarray(labels['unl_d'], c_unl_d.dtype)
    st['last_logits'] = {'unl': c_unl, 'unl_d': c_unl_d}
    X_P = np.concatenate([b['x_l_d'], b['x_u_d']], axis=0)
    Y_P = np.concatenate([b['y_l_d'], oh_unl_d], axis=0)
    grads, total = {}, 0.0
    for key, img, y, target, wgt in (('D_real', X_P, Y_P, 1.0, 1.0), ('D_fake', Gimg.reshape((-1,) + X_P.shape[1:]), b['y_g'], 0.0, 0.5),
                                     ('D_unl', b['x_u_c'], oh_unl, 0.0, 0.5)):
        logits, caches, _ = N.seq_fwd(P, DL, img, y, rnd[key], True)
        l, dl = T.bce_mean(logits, np.full_like(logits, target))
        total += wgt * l
        g, _ = N.seq_bwd(P, DL, caches, (wgt * dl).astype(logits.dtype), y, rnd[key])
        for k, v in g.items():
            grads[k] = grads.get(k, 0) + v
    _adam(st, 'D', grads, hyper['lr'], hyper['beta1'])
    return float(total)


def g_phase(st, data, b, rnd, hyper):
    P = st['P']
    GL, DL = N.generator_layers(data), N.discriminator_layers(data)
    bnu = {}
    Gimg, gc, _ = N.seq_fwd(P, GL, b['z_g'], b['y_g'], {}, True, bnu)
    N.commit_bn(P, bnu)
    img = Gimg.reshape((-1,) + N.image_shape(data))
    logits, dc, _ = N.seq_fwd(P, DL, img, b['y_g'], rnd['D_fake'], True)
    l, dl = T.bce_mean(logits, np.ones_like(logits))
    _, dimg = N.seq_bwd(P, DL, dc, (0.5 * dl).astype(logits.dtype), b['y_g'], rnd['D_fake'], want_params=False)
    grads, _ = N.seq_bwd(P, GL, gc, dimg.reshape(Gimg.shape), b['y_g'], {})
    _adam(st, 'G', grads, hyper['lr'], hyper['beta1'])
    return float(0.5 * l)


def c_phase(st, data, b, rnd, hyper):
    P = st['P']
    CL, DL = N.classifier_layers(data), N.discriminator_layers(data)
    bnu = {}
    Gimg, _, _ = _gen(P, data, b, bnu)
    gimg = Gimg.reshape((-1,) + N.image_shape(data))
    c_real, cc_real, _ = N.seq_fwd(P, CL, b['x_l_c'], None, rnd['C_real'], True, bnu)
    c_unl, cc_unl, _ = N.seq_fwd(P, CL, b['x_u_c'], None, rnd['C_unl'], True, bnu)
    c_fake, cc_fake, _ = N.seq_fwd(P, CL, gimg, None, rnd['C_fake'], True, bnu)
    N.commit_bn(P, bnu)
    oh = T.argmax_onehot(c_unl)
    d_unl, _, _ = N.seq_fwd(P, DL, b['x_u_c'], oh, rnd['D_unl'], True)
    lam1 = hyper['lambda_1']
    l_real, g_real = T.softmax_ce_mean(c_real, b['y_l_c'])
    l_fake, g_fake = T.softmax_ce_mean(c_fake, b['y_g'])
    l_unl, g_unl = T.c_unl_loss(c_unl, d_unl)
    l_ent, g_ent = T.entropy(c_unl)
    l_bal, g_bal = T.balance_entropy(c_unl)
    loss = 0.005 * l_unl + l_real + 1e-6 * l_ent + 1e-3 * l_bal + lam1 * l_fake
    f = c_real.dtype.type
    grads = {}
    for caches, dl, key in ((cc_real, g_real, 'C_real'), (cc_unl, f(0.005) * g_unl + f(1e-6) * g_ent + f(1e-3) * g_bal, 'C_unl'),
                            (cc_fake, f(lam1) * g_fake, 'C_fake')):
        g, _ = N.seq_bwd(P, CL, caches, dl.astype(c_real.dtype), None, rnd[key])
        for k, v in g.items():
            grads[k] = grads.get(k, 0) + v
    _adam(st, 'C', grads, hyper['cla_lr'], 0.5)
    for k in st['ema']:
        st['ema'][k] = T.ema_update(st['ema'][k], P[k])
    return float(loss)


def train_step(st, data, b, rnd, hyper):
    return d_phase(st, data, b, rnd['D'], hyper), g_phase(st, data, b, rnd['G'], hyper), c_phase(st, data, b, rnd['C'], hyper)


SIZES = {'mnist': dict(B_G=100, L_C=100, U_C=100, L_D=20, U_D=80), 'svhn': dict(B_G=100, L_C=50, U_C=50, L_D=20, U_D=80)}


def synth_batch(data, seed, sizes, dtype=np.float32):
    rng = np.random.default_rng(seed)
    shp = N.image_shape(data)
    lo = 0.0 if data == 'mnist' else -1.0
    proto = np.random.default_rng(1234).uniform(lo, 1, (10,) + shp)

    def imgs(n):
        y = rng.integers(0, 10, n)
        return np.clip(proto[y] + 0.25 * rng.standard_normal((n,) + shp), lo, 1).astype(dtype), np.eye(10, dtype=dtype)[y]
    b = {}
    b['x_l_c'], b['y_l_c'] = imgs(sizes['L_C'])
    b['x_l_d'], b['y_l_d'] = imgs(sizes['L_D'])
    xu, _ = imgs(sizes['U_D'] + sizes['U_C'])
    b['x_u_d'], b['x_u_c'] = xu[:sizes['U_D']], xu[sizes['U_D']:]
    b['z_g'] = rng.uniform(-1, 1, (sizes['B_G'], 100)).astype(dtype)
    b['y_g'] = np.eye(10, dtype=dtype)[rng.integers(0, 10, sizes['B_G'])]
    return b


def synth_rnd(data, seed, sizes, dtype=np.float32):
    rng = np.random.default_rng(seed)
    h, w, c = N.image_shape(data)

    def c_rnd(n):
        if data == 'mnist':
            return {'noise': (0.3 * rng.standard_normal((n, 28, 28, 1))).astype(dtype),
                    'drop1': (rng.random((n, 14, 14, 32)) < 0.5).astype(dtype), 'drop2': (rng.random((n, 7, 7, 64)) < 0.5).astype(dtype)}
        return {'drop0': (rng.random((n, 32, 32, 3)) < 0.8).astype(dtype), 'drop1': (rng.random((n, 16, 16, 128)) < 0.5).astype(dtype),
                'drop2': (rng.random((n, 8, 8, 256)) < 0.5).astype(dtype)}

    def d_rnd(n):
        if data == 'mnist':
            widths = (784, 1000, 500, 250, 250, 250)
            return {'noise%d' % i: (0.2 * rng.standard_normal((n, wd))).astype(dtype) for i, wd in enumerate(widths)}
        return {'drop0': (rng.random((n, 32, 32, 3)) < 0.8).astype(dtype), 'drop1': (rng.random((n, 16, 16, 32)) < 0.8).astype(dtype),
                'drop2': (rng.random((n, 8, 8, 64)) < 0.8).astype(dtype)}
    s = sizes
    return {'D': {'C_unl': c_rnd(s['U_C']), 'C_unl_d': c_rnd(s['U_D']), 'D_real': d_rnd(s['L_D'] + s['U_D']), 'D_fake': d_rnd(s['B_G']),
                  'D_unl': d_rnd(s['U_C'])},
            'G': {'D_fake': d_rnd(s['B_G'])},
            'C': {'C_real': c_rnd(s['L_C']), 'C_unl': c_rnd(s['U_C']), 'C_fake': c_rnd(s['B_G']), 'D_unl': d_rnd(s['U_C'])}}
