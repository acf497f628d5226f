"""ORACLE — test infrastructure only (see oracle/tf_ops.py header; PARITY UNPINNED).

CPU restatement of ONE Triple-GAN training iteration on the CIFAR-10 config:
the three sequential solver runs of Training/Train_goodGAN.py:266-276
(D-update, G-update, C-update + EMA), each executing only the sub-graph its
fetches need, with `_loss_GAN` (Training/train_base.py:113-154),
TF-form Adam (train_base.py:91-97; beta1 0.5, Train_goodGAN.py:85-87) and
EMA(0.9999) of the classifier variables (Train_goodGAN.py:101-103).

Deterministic order of the racing pop_mean updates: call-site order
of forward_pass (Model/Good_GAN_cifar10.py:228-240) — C_real, C_unl, C_unl_rep,
C_unl_d, C_fake — restricted to the applications a solver run executes.
"""
import numpy as np
from . import tf_ops as T
from . import nets_cifar10 as N


def trainable(names):
    return [n for n in names if 'pop_mean' not in n and 'moving_' not in n]


def new_state(P):
    """P: dict name -> array for all three nets (pop_mean included)."""
    st = {'P': {k: v.copy() for k, v in P.items()}, 't': {'D': 0, 'G': 0, 'C': 0}}
    tr = trainable(P.keys())
    st['m'] = {k: np.zeros_like(P[k]) for k in tr}
    st['v'] = {k: np.zeros_like(P[k]) for k in tr}
    st['ema'] = {k: P[k].copy() for k in tr if k.startswith('classifier/')}
    return st


def _adam(st, net, grads, lr, beta1):
    st['t'][net] += 1
    t = st['t'][net]
    st.setdefault('last_grads', {})[net] = grads          # kept for the parity tests (pre-Adam comparison)
    for k, g in grads.items():
        st['P'][k], st['m'][k], st['v'][k] = T.adam_update(
            st['P'][k], g.astype(st['P'][k].dtype), st['m'][k], st['v'][k], t, lr, beta1)


def _commit_pop(P, pop_updates):
    for p, v in pop_updates.items():
        P[p + 'meanOnlyBatchNormalization/pop_mean'] = v


def d_phase(st, batch, rnd, hyper, zca):
    """sess.run([d_solver, d_loss]) — Train_goodGAN.py:267."""
    P = st['P']
    Gimg, _ = N.generator_fwd(P, batch['z_g'], batch['y_g'], moving=P)
    pops = {}
    c_unl, _, _ = N.classifier_fwd(P, N.zca_apply(batch['x_u_c'], *zca), True, rnd['C_unl'], pops)
    c_unl_d, _, _ = N.classifier_fwd(P, N.zca_apply(batch['x_u_d'], *zca), True, rnd['C_unl_d'], pops)
    _commit_pop(P, pops)
    X_P = np.concatenate([batch['x_l_d'], batch['x_u_d']], axis=0)
    Y_P = np.concatenate([batch['y_l_d'], T.argmax_onehot(c_unl_d)], axis=0)
    grads = {}
    total = 0.0
    for key, img, y, target, wgt in (
            ('D_real', X_P, Y_P, 1.0, 1.0),
            ('D_fake', Gimg, batch['y_g'], 0.0, 0.5),
            ('D_unl', batch['x_u_c'], T.argmax_onehot(c_unl), 0.0, 0.5)):
        logits, c = N.discriminator_fwd(P, img, y, rnd[key])
        l, dl = T.bce_mean(logits, np.full_like(logits, target))
        total += wgt * l
        g, _ = N.discriminator_bwd(P, c, (wgt * dl).astype(logits.dtype), rnd[key])
        for k, v in g.items():
            grads[k] = grads.get(k, 0) + v
    _adam(st, 'D', grads, hyper['lr'], hyper['beta1'])
    return float(total)


def g_phase(st, batch, rnd, hyper):
    """sess.run([g_solver, g_loss]) — Train_goodGAN.py:270."""
    P = st['P']
    Gimg, gc = N.generator_fwd(P, batch['z_g'], batch['y_g'], moving=P)
    logits, c = N.discriminator_fwd(P, Gimg, batch['y_g'], rnd['D_fake'])
    l, dl = T.bce_mean(logits, np.ones_like(logits))
    _, dimg = N.discriminator_bwd(P, c, (0.5 * dl).astype(logits.dtype), rnd['D_fake'],
                                  want_weight_grads=False, want_input_grad=True)
    grads = N.generator_bwd(P, gc, dimg)
    _adam(st, 'G', grads, hyper['lr'], hyper['beta1'])
    return float(0.5 * l)


def c_phase(st, batch, rnd, hyper, zca):
    """sess.run([c_solver, c_loss]) — Train_goodGAN.py:275 (c_solver includes the EMA apply)."""
    P = st['P']
    Gimg, _ = N.generator_fwd(P, batch['z_g'], batch['y_g'], moving=P)
    pops = {}
    x_u_c_z = N.zca_apply(batch['x_u_c'], *zca)
    c_real, _, cc_real = N.classifier_fwd(P, N.zca_apply(batch['x_l_c'], *zca), True, rnd['C_real'], pops)
    c_unl, _, cc_unl = N.classifier_fwd(P, x_u_c_z, True, rnd['C_unl'], pops)
    c_rep, _, cc_rep = N.classifier_fwd(P, x_u_c_z, True, rnd['C_unl_rep'], pops)
    c_fake, _, cc_fake = N.classifier_fwd(P, N.zca_apply(Gimg, *zca), True, rnd['C_fake'], pops)
    _commit_pop(P, pops)
    d_unl, _ = N.discriminator_fwd(P, batch['x_u_c'], T.argmax_onehot(c_unl), rnd['D_unl'])

    lam1, lam2 = hyper['lambda_1'], hyper['lambda_2']
    l_real, g_real = T.softmax_ce_mean(c_real, batch['y_l_c'])
    l_fake, g_fake = T.softmax_ce_mean(c_fake, batch['y_g'])
    l_unl, g_unl = T.c_unl_loss(c_unl, d_unl)
    l_ent, g_ent = T.entropy(c_unl)
    l_bal, g_bal = T.balance_entropy(c_unl)
    l_mse, g_mse_unl, g_mse_rep = T.mse_mean(c_unl, c_rep)
    loss = 0.01 * 0.5 * l_unl + (l_real + 1e-6 * l_ent + 1e-3 * l_bal) + lam1 * l_fake + lam2 * l_mse
    f = c_real.dtype.type
    d_unl_logits = f(0.005) * g_unl + f(1e-6) * g_ent + f(1e-3) * g_bal + f(lam2) * g_mse_unl
    grads = {}
    for cache, dl, key in ((cc_real, g_real, 'C_real'), (cc_unl, d_unl_logits, 'C_unl'),
                           (cc_rep, f(lam2) * g_mse_rep, 'C_unl_rep'), (cc_fake, f(lam1) * g_fake, 'C_fake')):
        g = N.classifier_bwd(P, cache, dl.astype(c_real.dtype), rnd[key])
        for k, v in g.items():
            grads[k] = grads.get(k, 0) + v
    _adam(st, 'C', grads, hyper['cla_lr'], 0.5)
    for k in st['ema']:
        st['ema'][k] = T.ema_update(st['ema'][k], P[k])
    return float(loss)


def train_step(st, batch, rnd, hyper, zca):
    """One iteration: D -> G -> C.  rnd = {'D': {...}, 'G': {...}, 'C': {...}}."""
    d = d_phase(st, batch, rnd['D'], hyper, zca)
    g = g_phase(st, batch, rnd['G'], hyper)
    c = c_phase(st, batch, rnd['C'], hyper, zca)
    return d, g, c


# ------------------------------------------------------------------ helpers shared by tests/bench

def init_params(seed=0, dtype=np.float32):
    """Initial values per SURVEY App. A.1/C.6: C: V~N(0,.05^2), g=1, b=0, pop_mean=0;
    G/D: He (variance_scaling factor 2, FAN_IN, truncated normal, std sqrt(1.3*2/fan_in)
    with fan_in = shape[-2]*receptive field), bias 0, BN gamma 1 / beta 0."""
    rng = np.random.default_rng(seed)
    P = {}

    def he(shape):
        fan_in = shape[-2] * int(np.prod(shape[:-2])) if len(shape) > 2 else shape[0]
        std = np.sqrt(1.3 * 2.0 / fan_in)
        x = rng.standard_normal(shape)
        bad = np.abs(x) > 2
        while bad.any():
            x[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(x) > 2
        return (x * std).astype(dtype)

    for name, shape in N.generator_param_shapes() + N.discriminator_param_shapes():
        if name.endswith('kernel'):
            P[name] = he(shape)
        elif name.endswith('gamma'):
            P[name] = np.ones(shape, dtype)
        else:
            P[name] = np.zeros(shape, dtype)
    for name, shape in N.generator_moving_shapes():          # contrib batch_norm: moving_mean 0, moving_variance 1 (no random draw)
        P[name] = (np.ones if name.endswith('variance') else np.zeros)(shape, dtype)
    for name, shape in N.classifier_param_shapes():
        if name.endswith('/V'):
            P[name] = (rng.standard_normal(shape) * 0.05).astype(dtype)
        elif name.endswith('/g'):
            P[name] = np.ones(shape, dtype)
        else:
            P[name] = np.zeros(shape, dtype)
    return P


SIZES = dict(B_G=100, L_C=50, U_C=50, L_D=20, U_D=80)


def synth_batch(seed, sizes=SIZES, dtype=np.float32, noise=0.25, mix=0.0):
    """SURVEY §8d synthetic CIFAR-shaped batch: class-prototype images in [-1,1]; `noise` = pixel-noise sigma of the task.
    mix > 0 (the non-saturating long-horizon fixture only, tests/golden/make_golden_long.py): every image is a blend
    a*proto[y] + (1-a)*proto[y'] of its own class with a random OTHER class, a ~ U(1-mix, 1) — with mix = 0.5 images near a = 0.5 are
    genuinely ambiguous, so the error rate has a floor above zero and depends on how well the decision boundaries are learnt."""
    rng = np.random.default_rng(seed)
    proto = np.random.default_rng(1234).uniform(-1, 1, (10, 32, 32, 3))

    def imgs(n):
        y = rng.integers(0, 10, n)
        base = proto[y]
        if mix > 0:
            other = (y + rng.integers(1, 10, n)) % 10
            a = (1.0 - mix * rng.random(n)).reshape(n, 1, 1, 1)
            base = a * base + (1.0 - a) * proto[other]
        x = np.clip(base + noise * rng.standard_normal((n, 32, 32, 3)), -1, 1)
        return x.astype(dtype), np.eye(10, dtype=dtype)[y]

    b = {}
    b['x_l_c'], b['y_l_c'] = imgs(sizes['L_C'])
    b['x_l_d'], b['y_l_d'] = imgs(sizes['L_D'])
    xu, _ = imgs(sizes['U_D'] + sizes['U_C'])
    b['x_u_d'], b['x_u_c'] = xu[:sizes['U_D']], xu[sizes['U_D']:]   # Train_goodGAN.py:255-256
    b['z_g'] = rng.uniform(-1, 1, (sizes['B_G'], 100)).astype(dtype)
    b['y_g'] = np.eye(10, dtype=dtype)[rng.integers(0, 10, sizes['B_G'])]
    return b


def synth_zca(seed=4321, dim=3072, dtype=np.float32):
    """mean 0 + seeded random orthogonal matrix (real constants are absent)."""
    q, _ = np.linalg.qr(np.random.default_rng(seed).standard_normal((dim, dim)))
    return np.zeros(dim, dtype), q.astype(dtype)


def synth_rnd(seed, sizes=SIZES, dtype=np.float32):
    """All dropout keep-masks and Gaussian noises of one iteration."""
    rng = np.random.default_rng(seed)

    def c_rnd(n):
        return {'noise': (0.15 * rng.standard_normal((n, 32, 32, 3))).astype(dtype),
                'drop1': (rng.random((n, 16, 16, 128)) < 0.5).astype(dtype),
                'drop2': (rng.random((n, 8, 8, 256)) < 0.5).astype(dtype)}

    def d_rnd(n):
        return {'drop0': (rng.random((n, 32, 32, 3)) < 0.8).astype(dtype),
                'drop1': (rng.random((n, 16, 16, 32)) < 0.8).astype(dtype),
                'drop2': (rng.random((n, 8, 8, 64)) < 0.8).astype(dtype)}

    s = sizes
    return {
        'D': {'C_unl': c_rnd(s['U_C']), 'C_unl_d': c_rnd(s['U_D']),
              'D_real': d_rnd(s['L_D'] + s['U_D']), 'D_fake': d_rnd(s['B_G']), 'D_unl': d_rnd(s['U_C'])},
        'G': {'D_fake': d_rnd(s['B_G'])},
        'C': {'C_real': c_rnd(s['L_C']), 'C_unl': c_rnd(s['U_C']), 'C_unl_rep': c_rnd(s['U_C']),
              'C_fake': c_rnd(s['B_G']), 'D_unl': d_rnd(s['U_C'])},
    }
