"""Oracle for Model/Good_GAN.py (MNIST, SVHN): parameter counts against SURVEY App. A.2 and whole-network gradients
against torch autograd of a torch evaluation of the same layer lists (float64)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets_goodgan as N
from oracle import step_goodgan as S
from test_oracle_ops import tconv

torch.set_num_threads(4)


def tt(x, grad=False):
    return torch.tensor(np.asarray(x, np.float64), requires_grad=grad)


def wn(v, g, axis=-1):
    axis %= v.dim()
    axes = [a for a in range(v.dim()) if a != axis]
    shp = [1] * v.dim()
    shp[axis] = -1
    return g.reshape(shp) * v / torch.sqrt((v * v).sum(dim=axes, keepdim=True))


def t_deconv(x, w):
    n, h = x.shape[0], x.shape[1]
    X = torch.zeros((n, 2 * h, 2 * h, w.shape[2]), dtype=torch.float64, requires_grad=True)
    (out,) = torch.autograd.grad(tconv(X, w, 2, 'SAME'), X, x, create_graph=True)
    return out


def t_seq(P, layers, x, y, rnd, train=True):
    feat = None
    for l in layers:
        k = l[0]
        if k == 'concat_y':
            x = torch.cat([x, y], 1)
        elif k == 'cond_concat':
            n, h, w, _ = x.shape
            x = torch.cat([x, y.reshape(n, 1, 1, -1).expand(n, h, w, y.shape[1])], 3)
        elif k == 'reshape':
            x = x.reshape((x.shape[0],) + tuple(l[1]))
        elif k == 'dense':
            x = x @ P[l[1] + '/kernel'] + P[l[1] + '/bias']
        elif k in ('wn_dense', 'nin'):
            x = x @ wn(P[l[1] + '/V'], P[l[1] + '/g']) + P[l[1] + '/b']
        elif k == 'conv':
            x = tconv(x, P[l[1] + '/kernel'], l[3], 'SAME') + P[l[1] + '/bias']
        elif k == 'wn_conv':
            x = tconv(x, wn(P[l[1] + '/V'], P[l[1] + '/g']), l[3], 'SAME') + P[l[1] + '/b']
        elif k == 'deconv':
            x = t_deconv(x, P[l[1] + '/kernel']) + P[l[1] + '/bias']
        elif k == 'wn_deconv':
            x = t_deconv(x, wn(P[l[1] + '/V'], P[l[1] + '/g'], 2)) + P[l[1] + '/b']
        elif k == 'act':
            x = {'relu': torch.relu, 'lrelu': lambda v: F.leaky_relu(v, 0.2), 'softplus': F.softplus, 'sigmoid': torch.sigmoid,
                 'tanh': torch.tanh}[l[1]](x)
        elif k == 'bn':
            axes = list(range(x.dim() - 1))
            if train:
                mu = x.mean(dim=axes)
                var = ((x - mu) ** 2).mean(dim=axes)
            else:
                mu, var = P[l[1] + '/moving_mean'], P[l[1] + '/moving_variance']
            x = P[l[1] + '/gamma'] * (x - mu) / torch.sqrt(var + 1e-5) + P[l[1] + '/beta']
        elif k == 'noise':
            x = x + tt(rnd[l[1]])
        elif k == 'dropout':
            if l[3] or train:
                x = x * tt(rnd[l[1]]) / (1 - l[2])
        elif k == 'maxpool':
            x = F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        elif k == 'gmean':
            x = x.mean(dim=(1, 2))
        elif k == 'feature':
            feat = x
    return x, feat


def scrambled(data, seed):
    P = N.init_params(data, seed, np.float64)
    rng = np.random.default_rng(seed + 1)
    for k in P:
        if k.endswith(('kernel',)):
            P[k] = P[k] * 0.05                      # the reference's mean-.02/std-1 initialiser saturates everything
        elif k.endswith(('/g', 'gamma')):
            P[k] = 1 + 0.3 * rng.standard_normal(P[k].shape)
        elif k.endswith(('/b', 'bias', 'beta')):
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
        elif k.endswith('moving_variance'):
            P[k] = 0.5 + rng.random(P[k].shape)
        elif k.endswith('moving_mean'):
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    return P


def test_parameter_counts_match_survey():
    def count(data, net):
        return sum(int(np.prod(s)) for n, s, _ in N.param_shapes(data) if n.startswith(net) and 'moving_' not in n)
    assert count('mnist', 'good_generator') == 714408 and count('mnist', 'discriminator') == 1561262
    assert count('mnist', 'classifier') == 279326
    assert count('svhn', 'discriminator') == 339436 and count('svhn', 'classifier') == 3124254


@pytest.mark.parametrize("data", ['mnist', 'svhn'])
@pytest.mark.parametrize("net", ['G', 'D', 'C'])
def test_network_gradients(data, net):
    P = scrambled(data, 3)
    n = 3
    sizes = dict(B_G=n, L_C=n, U_C=n, L_D=1, U_D=n - 1)
    b = S.synth_batch(data, 5, sizes, np.float64)
    rnd = S.synth_rnd(data, 6, sizes, np.float64)
    if net == 'G':
        layers, x, y, r = N.generator_layers(data), b['z_g'], b['y_g'], {}
    elif net == 'D':
        layers, x, y, r = N.discriminator_layers(data), b['x_l_c'], b['y_l_c'], rnd['G']['D_fake']
    else:
        layers, x, y, r = N.classifier_layers(data), b['x_l_c'], None, rnd['C']['C_real']
    out, caches, feat = N.seq_fwd(P, layers, x, y, r, True)
    names = [k for k in P if k.startswith({'G': 'good_generator', 'D': 'discriminator', 'C': 'classifier'}[net]) and 'moving_' not in k]
    TP = {k: tt(v, k in names) for k, v in P.items()}
    tx = tt(x, True)
    to, tf = t_seq(TP, layers, tx, None if y is None else tt(y), r)
    np.testing.assert_allclose(out, to.detach().numpy(), rtol=1e-8, atol=1e-10)
    rng = np.random.default_rng(7)
    do = rng.standard_normal(out.shape)
    loss = (to * tt(do)).sum()
    df = None
    if feat is not None:
        df = rng.standard_normal(feat.shape)
        loss = loss + (tf * tt(df)).sum()
    loss.backward()
    G, dx = N.seq_bwd(P, layers, caches, do, y, r, dfeat=df)
    assert set(G) == set(names)
    gmax = max(float(TP[k].grad.abs().max()) for k in names)
    for k in names:
        ref = TP[k].grad.numpy()
        sc = max(np.abs(ref).max(), 1e-6 * gmax)    # a bias in front of a batch norm has an analytically zero gradient: pure noise
        np.testing.assert_allclose(G[k] / sc, ref / sc, rtol=1e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-6, atol=1e-10)


def test_eval_mode_uses_moving_statistics_and_updates_chain():
    P = scrambled('mnist', 4)
    b = S.synth_batch('mnist', 8, dict(B_G=2, L_C=2, U_C=2, L_D=1, U_D=1), np.float64)
    rnd = S.synth_rnd('mnist', 9, dict(B_G=2, L_C=2, U_C=2, L_D=1, U_D=1), np.float64)['C']['C_real']
    L = N.classifier_layers('mnist')
    out, _, _ = N.seq_fwd(P, L, b['x_l_c'], None, rnd, False)
    TP = {k: tt(v) for k, v in P.items()}
    to, _ = t_seq(TP, L, tt(b['x_l_c']), None, rnd, train=False)
    np.testing.assert_allclose(out, to.numpy(), rtol=1e-9, atol=1e-11)
    bnu = {}
    N.seq_fwd(P, L, b['x_l_c'], None, rnd, True, bnu)
    first = {k: (a.copy(), v.copy()) for k, (a, v) in bnu.items()}
    N.seq_fwd(P, L, b['x_l_c'], None, rnd, True, bnu)
    k = 'classifier/c_h0_bn0'
    mu = (first[k][0] - 0.9 * P[k + '/moving_mean']) / 0.1
    np.testing.assert_allclose(bnu[k][0], 0.9 * first[k][0] + 0.1 * mu, rtol=1e-10)       # second application chains on the first


def test_bf16_rounding_amplifies_accumulation_noise():
    """Evidence for the tolerances of the bf16 MFMA configuration (tests/test_gpu_goodgan.py, 'svhn-bf16'): the SAME
    algorithm with the SAME operand-rounding rule (oracle/tf_ops.py MFMA_BF16), accumulated once in float32 and once in
    float64, agrees to ~5e-6 without the rounding and only to ~1e-3..1e-2 with it — an activation that differs by fp32
    noise lands on the other side of a bf16 rounding boundary and then differs by 2^-8.  No implementation pair can be
    closer than this, so this is the floor of the HIP-vs-oracle comparison in that configuration."""
    from oracle import tf_ops as T
    data = 'svhn'
    P32 = {k: v.astype(np.float32) for k, v in scrambled(data, 3).items()}
    P64 = {k: v.astype(np.float64) for k, v in P32.items()}
    sizes = dict(B_G=5, L_C=5, U_C=5, L_D=1, U_D=4)
    b = S.synth_batch(data, 5, sizes)
    rel = lambda a, r: np.abs(a - r).max() / np.abs(r).max()
    GL = N.generator_layers(data)
    out = {}
    try:
        for mode in (False, True):
            T.MFMA_BF16 = mode
            o32, c32, _ = N.seq_fwd(P32, GL, b['z_g'], b['y_g'], {}, True)
            o64, c64, _ = N.seq_fwd(P64, GL, b['z_g'].astype(np.float64), b['y_g'].astype(np.float64), {}, True)
            do = np.random.default_rng(7).standard_normal(o64.shape)
            g32, _ = N.seq_bwd(P32, GL, c32, do.astype(np.float32), b['y_g'], {})
            g64, _ = N.seq_bwd(P64, GL, c64, do, b['y_g'].astype(np.float64), {})
            gerr = max(np.linalg.norm(g32[k] - g64[k]) / np.linalg.norm(g64[k]) for k in g64 if np.linalg.norm(g64[k]) > 1e-12)
            out[mode] = (rel(o32, o64), gerr)
    finally:
        T.MFMA_BF16 = False
    assert out[False][0] < 1e-4 and out[False][1] < 1e-4, out
    assert out[True][0] > 1e-4 and out[True][1] > 1e-4, out          # the rounding step is what amplifies
    assert out[True][0] < 3e-2 and out[True][1] < 0.35, out           # and stays inside the bf16 tolerances used on the GPU
    # classifier (ten batch norms deep): the floor is percent-level for the gradients
    CL = N.classifier_layers(data)
    rnd = S.synth_rnd(data, 6, sizes)['C']['C_real']
    r64 = {k: v.astype(np.float64) for k, v in rnd.items()}
    try:
        T.MFMA_BF16 = True
        l32, c32, _ = N.seq_fwd(P32, CL, b['x_l_c'], None, rnd, True, {})
        l64, c64, _ = N.seq_fwd(P64, CL, b['x_l_c'].astype(np.float64), None, r64, True, {})
        dl = np.random.default_rng(7).standard_normal(l64.shape)
        g32, _ = N.seq_bwd(P32, CL, c32, dl.astype(np.float32), None, rnd)
        g64, _ = N.seq_bwd(P64, CL, c64, dl, None, r64)
    finally:
        T.MFMA_BF16 = False
    agg = np.sqrt(sum(np.linalg.norm(g32[k] - g64[k]) ** 2 for k in g64)) / np.sqrt(sum(np.linalg.norm(g64[k]) ** 2 for k in g64))
    assert 1e-3 < rel(l32, l64) < 3e-2, rel(l32, l64)
    assert 1e-2 < agg < 0.35, agg
