"""Whole-network gradient check of the oracle's hand-derived backward against torch
autograd of an independent torch restatement (float64, tiny batch)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import nets_cifar10 as N
from oracle import step_cifar10 as S
from oracle import tf_ops as T
from test_oracle_ops import tconv

torch.set_num_threads(4)
D64 = np.float64


def tt(x, grad=False):
    return torch.tensor(np.asarray(x, D64), requires_grad=grad)


def lrelu(x):
    return torch.relu(x) - 0.2 * torch.relu(-x)


def wn(v, g):
    axes = list(range(v.dim() - 1))
    return g * v / torch.sqrt((v * v).sum(dim=axes, keepdim=True))


def t_classifier(P, x, rnd):
    x = x + tt(rnd['noise'])
    for name, cout, pad in N.C_CONVS:
        p = 'classifier/%s/' % name
        x = tconv(x, wn(P[p + 'V'], P[p + 'g']), 1, pad)
        x = lrelu(x - x.mean(dim=(0, 1, 2)) + P[p + 'b'])
        if name in N.C_POOL_AFTER:
            x = F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            x = x * tt(rnd[N.C_POOL_AFTER[name]]) / 0.5
    for name in ('NiN1', 'NiN2'):
        p = 'classifier/%s/%s/' % (name, name)
        s = x.shape
        x2 = x.reshape(-1, s[-1]) @ P[p + 'V'] * (P[p + 'g'] / torch.sqrt((P[p + 'V'] ** 2).sum(0)))
        x = lrelu(x2 - x2.mean(0) + P[p + 'b']).reshape(s[0], s[1], s[2], -1)
    feat = x.amax(dim=(1, 2))
    p = 'classifier/output_dense/'
    x2 = feat @ P[p + 'V'] * (P[p + 'g'] / torch.sqrt((P[p + 'V'] ** 2).sum(0)))
    return x2 - x2.mean(0) + P[p + 'b'], feat


def bn(x, g, b):
    axes = list(range(x.dim() - 1))
    mu = x.mean(dim=axes)
    var = ((x - mu) ** 2).mean(dim=axes)
    return g * (x - mu) / torch.sqrt(var + 1e-5) + b


def cc(x, y):
    n, h, w, _ = x.shape
    return torch.cat([x, y.reshape(n, 1, 1, -1).expand(n, h, w, y.shape[1])], dim=3)


def t_deconv(x, w):
    n, h, _, _ = x.shape
    X = torch.zeros((n, 2 * h, 2 * h, w.shape[2]), dtype=torch.float64, requires_grad=True)
    (out,) = torch.autograd.grad(tconv(X, w, 2, 'SAME'), X, x, create_graph=True)
    return out


def t_generator(P, z, y):
    g = 'good_generator/'
    h = torch.relu(torch.cat([z, y], 1) @ P[g + 'gg_h0_lin/gg_h0_lin/kernel'] + P[g + 'gg_h0_lin/gg_h0_lin/bias'])
    h = bn(h, P[g + 'gg_bn0/gamma'], P[g + 'gg_bn0/beta']).reshape(-1, 4, 4, 512)
    h = cc(h, y)
    for i, (name, cout) in enumerate(N.G_DECONVS):
        p = g + '%s/%s/' % (name, name)
        h = t_deconv(h, P[p + 'kernel']) + P[p + 'bias']
        if i < 2:
            h = cc(bn(torch.relu(h), P[g + 'gg_bn%d/gamma' % (i + 1)], P[g + 'gg_bn%d/beta' % (i + 1)]), y)
    return torch.tanh(h)


def t_discriminator(P, img, y, rnd):
    h = img * tt(rnd['drop0']) / 0.8
    for name, cout, s, drop in N.D_CONVS:
        p = 'discriminator/%s/%s/' % (name, name)
        h = lrelu(tconv(cc(h, y), P[p + 'kernel'], s, 'SAME') + P[p + 'bias'])
        if drop:
            h = h * tt(rnd[drop]) / 0.8
    h = torch.cat([h.mean(dim=(1, 2)), y], 1)
    return h @ P['discriminator/lin/lin/kernel'] + P['discriminator/lin/lin/bias']


def _params64(seed, scramble=True):
    P = S.init_params(seed, D64)
    if scramble:  # break the g=1,b=0 symmetry so every gradient path is exercised
        rng = np.random.default_rng(seed + 100)
        for k in P:
            if k.endswith(('/g', 'gamma')):
                P[k] = 1 + 0.3 * rng.standard_normal(P[k].shape)
            elif k.endswith(('/b', 'bias', 'beta')):
                P[k] = 0.1 * rng.standard_normal(P[k].shape)
    return P


def _cmp(G, TP, names, rtol=1e-6):
    for k in names:
        ref = TP[k].grad.numpy()
        scale = np.abs(ref).max() + 1e-30
        np.testing.assert_allclose(G[k] / scale, ref / scale, rtol=rtol, atol=1e-9, err_msg=k)


def test_classifier_grads():
    P = _params64(0)
    sizes = dict(S.SIZES, L_C=3)
    rnd = S.synth_rnd(1, sizes, D64)['C']['C_real']
    x = S.synth_batch(2, sizes, D64)['x_l_c']
    logits, feat, cache = N.classifier_fwd(P, x, True, rnd)
    TP = {k: tt(v, True) for k, v in P.items() if k.startswith('classifier/') and 'pop_mean' not in k}
    tl, tf_ = t_classifier(TP, tt(x), rnd)
    np.testing.assert_allclose(logits, tl.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(feat, tf_.detach().numpy(), rtol=1e-9, atol=1e-11)
    rng = np.random.default_rng(3)
    dl, df = rng.standard_normal(logits.shape), rng.standard_normal(feat.shape)
    (tl * tt(dl)).sum().add((tf_ * tt(df)).sum()).backward()
    G = N.classifier_bwd(P, cache, dl, rnd, dfeat=df)
    assert set(G) == set(TP)
    _cmp(G, TP, TP.keys())


def test_classifier_eval_and_pop_mean_chain():
    P = S.init_params(0, D64)
    sizes = dict(S.SIZES, L_C=2)
    rnd = S.synth_rnd(1, sizes, D64)['C']['C_real']
    x = S.synth_batch(2, sizes, D64)['x_l_c']
    pops = {}
    N.classifier_fwd(P, x, True, rnd, pops)
    first = {k: v.copy() for k, v in pops.items()}
    N.classifier_fwd(P, x, True, rnd, pops)
    for k in first:  # second application starts from the first one's update: 0.9*(0.1 m) + 0.1 m
        np.testing.assert_allclose(pops[k], 1.9 * first[k], rtol=1e-12)
    l_eval, _, _ = N.classifier_fwd(P, x, False, rnd)
    assert l_eval.shape == (2, 10)


def test_generator_grads():
    P = _params64(1)
    sizes = dict(S.SIZES, B_G=3)
    b = S.synth_batch(4, sizes, D64)
    out, c = N.generator_fwd(P, b['z_g'], b['y_g'])
    TP = {k: tt(v, True) for k, v in P.items() if k.startswith('good_generator/') and 'moving_' not in k}
    to = t_generator(TP, tt(b['z_g']), tt(b['y_g']))
    np.testing.assert_allclose(out, to.detach().numpy(), rtol=1e-8, atol=1e-10)
    do = np.random.default_rng(5).standard_normal(out.shape)
    (to * tt(do)).sum().backward()
    G = N.generator_bwd(P, c, do)
    assert set(G) == set(TP)
    _cmp(G, TP, TP.keys(), rtol=1e-5)


def test_discriminator_grads_and_input_grad():
    P = _params64(2)
    sizes = dict(S.SIZES, B_G=3)
    b = S.synth_batch(6, sizes, D64)
    rnd = S.synth_rnd(7, sizes, D64)['G']['D_fake']
    img = np.tanh(np.random.default_rng(8).standard_normal((3, 32, 32, 3)))
    logits, c = N.discriminator_fwd(P, img, b['y_g'], rnd)
    TP = {k: tt(v, True) for k, v in P.items() if k.startswith('discriminator/')}
    ti = tt(img, True)
    tl = t_discriminator(TP, ti, tt(b['y_g']), rnd)
    np.testing.assert_allclose(logits, tl.detach().numpy(), rtol=1e-9, atol=1e-11)
    dl = np.random.default_rng(9).standard_normal(logits.shape)
    (tl * tt(dl)).sum().backward()
    G, dimg = N.discriminator_bwd(P, c, dl, rnd, True, True)
    assert set(G) == set(TP)
    _cmp(G, TP, TP.keys())
    np.testing.assert_allclose(dimg, ti.grad.numpy(), rtol=1e-7, atol=1e-12)


def test_param_counts_match_survey():
    """G 5 129 201, D 327 467, C 3 121 812 trainable parameters."""
    cnt = lambda shapes: sum(int(np.prod(s)) for n, s in shapes if 'pop_mean' not in n)
    assert cnt(N.generator_param_shapes()) == 5129201
    assert cnt(N.discriminator_param_shapes()) == 327467
    assert cnt(N.classifier_param_shapes()) == 3121812
