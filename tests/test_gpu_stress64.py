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
