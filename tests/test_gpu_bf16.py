"""bf16 MFMA conv path on the CIFAR-10 model (Context.mfma_dtype = 'bf16'): integration check against the fp32 path of
the same build on identical weights, batches and Philox streams.  (Kernel-level parity of the tg_*_bf16 launches is in
tests/test_gpu_igemm.py and tests/test_gpu_kernels.py; step-level parity against the oracle's bf16 emulation in
tests/test_gpu_goodgan.py 'svhn-bf16'.)"""
import numpy as np
import pytest

import gpu_common as G

pytestmark = pytest.mark.gpu


def _run(prec, n_iter=2):
    import torch
    from oracle import step_cifar10 as S
    sizes = dict(B_G=12, L_C=8, U_C=8, L_D=4, U_D=8)
    tr = G.fresh_trainer(G.make_config(sizes, MFMA_DTYPE=prec, SEED=3))
    tr.set_hyper(3e-4, 3e-3, 0.3, 0.5)
    b = S.synth_batch(31, sizes)
    tr.feed(b)
    out = []
    for _ in range(n_iter):
        tr.sample_latent()
        tr.train_iteration()
        out.append(tr.losses())
    torch.cuda.synchronize()
    return out, {k: s.p.detach().cpu().numpy().copy() for k, s in tr.cx.stores.items()}


def test_cifar10_bf16_step_tracks_fp32():
    lf, pf = _run('f32')
    lb, pb = _run('bf16')
    assert lf != lb                                               # the bf16 launches really ran
    for a, b in zip(lf, lb):
        for x, y in zip(a, b):
            assert np.isfinite(y) and abs(x - y) <= 3e-2 * max(1.0, abs(x)), (lf, lb)
    for k in pf:
        assert np.isfinite(pb[k]).all()
        # two Adam steps move every weight by at most ~2*lr; the two runs must stay within that envelope of each other
        assert np.abs(pf[k] - pb[k]).max() <= 2.1 * 2 * 3e-3, k
