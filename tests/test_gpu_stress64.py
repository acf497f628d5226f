"""64x64 stress configuration (BASELINE.json configs[4]; build-defined, no parity target): the deeper G/D/C tables of
Model/Good_GAN_stress64.py run through the same kernels; checks shapes, finite losses and that parameters move."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu


def test_stress64_two_iterations():
    import torch
    from gpu_common import fresh_trainer
    import bench_config as stress64
    from Model.Good_GAN_stress64 import Good_GAN_stress64
    cfg = stress64.make_config()
    # a quarter of the batch keeps the test at a few hundred ms
    cfg.BATCH_SIZE_G = cfg.BATCH_SIZE = 64
    cfg.BATCH_SIZE_L_C = cfg.BATCH_SIZE_U_C = 32
    cfg.BATCH_SIZE_L_D, cfg.BATCH_SIZE_U_D = 13, 51
    cfg.USE_HIP_GRAPH = False
    tr = fresh_trainer(cfg, Model=Good_GAN_stress64)
    tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
    specs = Good_GAN_stress64.param_specs()
    assert sum(int(np.prod(s)) for _, s, t, _ in specs['good_generator'] if t) > 5.1e6
    rng = np.random.default_rng(0)
    img = lambda n: rng.uniform(-1, 1, (n, 64, 64, 3)).astype(np.float32)
    oh = lambda n: np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    tr.feed(dict(x_l_c=img(32), y_l_c=oh(32), x_l_d=img(13), y_l_d=oh(13), x_u_d=img(51), x_u_c=img(32)))
    before = {k: s.p.clone() for k, s in tr.cx.stores.items()}
    for _ in range(2):
        tr.sample_latent()
        tr.train_iteration()
    torch.cuda.synchronize()
    losses = tr.losses()
    assert all(np.isfinite(losses)) and all(v > 0 for v in losses), losses
    for k, s in tr.cx.stores.items():
        assert torch.isfinite(s.p).all()
        assert (s.p != before[k]).any(), k
    out = tr.sample(rng.uniform(-1, 1, (4, 100)).astype(np.float32), oh(4))
    assert out.shape == (4, 64, 64, 3) and np.abs(out).max() <= 1.0


FULL = dict(B_G=256, L_C=128, U_C=128, L_D=51, U_D=205)          # BASELINE.json configs[4] at its stated size


def _full_trainer(graph):
    from gpu_common import fresh_trainer
    import bench_config as stress64
    from Model.Good_GAN_stress64 import Good_GAN_stress64
    cfg = stress64.make_config()
    assert (cfg.BATCH_SIZE_G, cfg.BATCH_SIZE_L_C, cfg.BATCH_SIZE_U_C, cfg.BATCH_SIZE_L_D, cfg.BATCH_SIZE_U_D) == (256, 128, 128, 51, 205)
    cfg.USE_HIP_GRAPH = graph
    tr = fresh_trainer(cfg, Model=Good_GAN_stress64)
    tr.set_hyper(lambda_1=0.3, lambda_2=0.5)
    return tr


def _feed(tr, seed):
    rng = np.random.default_rng(seed)
    img = lambda n: rng.uniform(-1, 1, (n, 64, 64, 3)).astype(np.float32)
    oh = lambda n: np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    tr.feed(dict(x_l_c=img(FULL['L_C']), y_l_c=oh(FULL['L_C']), x_l_d=img(FULL['L_D']), y_l_d=oh(FULL['L_D']), x_u_d=img(FULL['U_D']),
                 x_u_c=img(FULL['U_C'])))


def test_stress64_at_its_stated_batch_sizes_properties():
    """bs 256/128/128/51/205, 64x64x3 (24 GB of activations).  No oracle runs at this size in test time; size-independent properties
    instead: (1) finite positive losses and every variable of the three networks moves; (2) hipGraph replay reproduces the eager run bit
    for bit over several iterations (the production mode against the plainly launched one: every buffer, statistic arena and RNG counter
    lines up); (3) applications batched into one classifier call are independent — the logits of [A | B] with per-application statistics
    equal those of A and B run alone; (4) ragged application sizes (51 / 205 rows, not multiples of any tile) leave the padding exactly zero."""
    import torch
    runs = {}
    for graph in (False, True):
        tr = _full_trainer(graph)
        before = {k: s.p.clone() for k, s in tr.cx.stores.items()}
        losses = []
        for it in range(4):                      # graph mode: eager, capture, replay, replay
            _feed(tr, 100 + it)
            tr.sample_latent()
            tr.train_iteration()
            losses.append(tr.losses())
        torch.cuda.synchronize()
        assert all(np.isfinite(l).all() and min(l) > 0 for l in map(np.asarray, losses)), losses
        for k, s in tr.cx.stores.items():
            assert torch.isfinite(s.p).all() and (s.p != before[k]).any(), k
        runs[graph] = (losses, {k: s.p.detach().cpu().numpy().copy() for k, s in tr.cx.stores.items()})
        if graph:
            assert all(g is not None for g in tr._graphs['full'])
    assert runs[True][0] == runs[False][0]
    for k in runs[False][1]:
        np.testing.assert_array_equal(runs[True][1][k], runs[False][1][k], err_msg=k)

    # (3) + (4): the classifier on [x_l_d (51) | x_u_d (205)] as two applications of one call vs alone, evaluation of the
    # statistics per application; dropout / noise injected so that the three calls see the same draws
    from tg.runtime import InjectedRNG
    cx, m = tr.cx, tr.model
    rng = np.random.default_rng(7)
    xa = rng.uniform(-1, 1, (51, 64, 64, 3)).astype(np.float32)
    xb = rng.uniform(-1, 1, (205, 64, 64, 3)).astype(np.float32)

    def draws(n, seed):
        r = np.random.default_rng(seed)
        with cx.phase_scope('probe', record=False):
            cx.rng = _Recorder(r)
            with cx.rng_scoped('probe/C'):
                m.classifier(cx.from_numpy(np.zeros((n, 64, 64, 3), np.float32)), True)
        return cx.rng.made

    class _Recorder(object):
        """records the (name, size) of every draw of one classifier application and answers with seeded numbers."""
        def __init__(self, r):
            self.r, self.made = r, {}

        def keep_mask(self, ctx_, name, n, keep):
            a = (self.r.random(n) < keep).astype(np.float32)
            self.made[name] = a
            return torch.from_numpy(a).to(cx.device)

        def normal(self, ctx_, name, n, std):
            a = (std * self.r.standard_normal(n)).astype(np.float32)
            self.made[name] = a
            return torch.from_numpy(a).to(cx.device)

        def advance(self, ctx_):
            pass

    da, db = draws(51, 1), draws(205, 2)
    outs = {}
    for tag, x, inj, segs in (('a', xa, da, None), ('b', xb, db, None),
                              ('ab', np.concatenate([xa, xb]), {k: np.concatenate([da[k], db[k]]) for k in da}, [51, 205])):
        cx.rng = InjectedRNG({'S%s/C/%s' % (tag, k): v for k, v in inj.items()}, cx.device)
        with cx.phase_scope('S' + tag, record=False):
            with cx.rng_scoped('S%s/C' % tag):
                logits, _ = m.classifier(cx.from_numpy(x), True, segments=segs)
            outs[tag] = logits.numpy()
            pad = logits.t.reshape(logits.n, logits.ld)[:, logits.c:]
            assert float(pad.abs().max()) == 0.0, tag
    ref = np.concatenate([outs['a'], outs['b']])
    assert np.abs(outs['ab'] - ref).max() <= 2e-5 * np.abs(ref).max(), np.abs(outs['ab'] - ref).max()
