"""HIP path against the committed golden vectors (tests/golden/): losses of ten free-running iterations, parameter
checksums, good_sampler output and evaluation-mode logits / accuracy on a fixed test split — without running the oracle."""
import os
import sys

import numpy as np
import pytest

import gpu_common as G

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
pytestmark = pytest.mark.gpu


def test_hip_path_matches_the_golden_vectors():
    import torch
    import make_golden as M                                   # seeds / sizes / input generators only
    from oracle import step_cifar10 as S
    from tg.runtime import InjectedRNG
    g = np.load(os.path.join(HERE, 'golden', 'cifar10_small_k10.npz'))
    tr = G.fresh_trainer(G.make_config(M.SIZES), S.init_params(0))
    tr.set_hyper(M.HYPER['lr'], M.HYPER['cla_lr'], M.HYPER['lambda_1'], M.HYPER['lambda_2'])
    cx = tr.cx
    z, y = M.sample_latents()
    xt, yt, noise = M.test_split()

    def evaluate():
        cx.rng = InjectedRNG({'val/C/noise': noise}, cx.device)
        acc = tr.evaluate([(xt, yt)])
        return acc

    # initial weights: sampler and evaluation are deterministic functions of the inputs -> fp32 tolerance
    assert G.rel_err(tr.sample(z, y), g['sample_init']) < 2e-4
    assert abs(evaluate() - float(g['acc_init'])) <= 1.0 / M.N_TEST + 1e-9        # one arg-max tie at most
    losses = []
    for k in range(M.K):
        b, r = M.inputs(k)
        cx.rng = InjectedRNG(G.injected_arrays(r), cx.device)
        tr.feed(b)
        tr.train_iteration(use_graph=False)
        losses.append(tr.losses())
    torch.cuda.synchronize()
    losses = np.asarray(losses)
    ref = g['losses']
    assert np.abs(losses[0] - ref[0]).max() <= 2e-4 * np.abs(ref[0]).max()          # first iteration: identical weights
    # afterwards the two trajectories drift (sign-like first Adam steps, tests/test_gpu_step.py docstring): bounded, not tight
    assert np.abs(losses - ref).max() <= 0.15, np.abs(losses - ref).max(axis=1)
    for net in ('good_generator', 'discriminator', 'classifier'):
        st = cx.stores[net]
        p = np.concatenate([st.get(k).reshape(-1).astype(np.float64) for k in st.names(True)])
        s1, s2 = g['checksum/' + net]
        lr = M.HYPER['cla_lr'] if net == 'classifier' else M.HYPER['lr']
        # free-running trajectories: elements whose gradients are rounding noise take +-lr steps of either sign, so the plain sum
        # random-walks by ~lr*K*sqrt(N) (allowed: 4x) between two correct implementations (a wrong step size or a missing update moves it by
        # ~lr*K*N, a thousand times more); the sum of squares is insensitive to that
        assert abs(p.sum() - s1) <= 4 * lr * M.K * np.sqrt(p.size) and abs((p * p).sum() - s2) <= 2e-3 * s2, net
    assert np.abs(tr.sample(z, y) - g['sample_final']).mean() <= 0.05
    assert abs(evaluate() - float(g['acc_final'])) <= 0.1


@pytest.mark.parametrize("data", ['mnist', 'svhn'])
def test_goodgan_hip_path_matches_the_golden_vectors(data):
    """Model/Good_GAN.py against tests/golden/goodgan_<data>_k5.npz."""
    import torch
    import make_golden_goodgan as M
    from Model.Good_GAN import Good_GAN
    from tg.runtime import InjectedRNG
    g = np.load(M.path(data))
    tr = G.fresh_trainer(G.make_config_goodgan(data, M.SIZES), M.init_params(data), Good_GAN)
    h = M.HYPER[data]
    tr.set_hyper(h['lr'], h['cla_lr'], h['lambda_1'], h['lambda_2'])
    cx, m = tr.cx, tr.model
    z, y = M.sample_latents()
    xt, yt, rnd = M.test_split(data)

    def evaluate():
        cx.rng = InjectedRNG({'val/C/' + k: v for k, v in rnd.items()}, cx.device)
        with cx.phase_scope('val', record=False):
            with cx.rng_scoped('val/C'):
                logits, _ = m.classifier(cx.from_numpy(xt), False)
        cx.rng = InjectedRNG({'val/C/' + k: v for k, v in rnd.items()}, cx.device)
        return logits.numpy(), tr.evaluate([(xt, yt)])

    def sample():
        return tr.sample(z, y).reshape(g['sample_init'].shape)

    # initial weights: deterministic functions of the inputs -> fp32 tolerance
    assert G.rel_err(sample(), g['sample_init']) < 2e-4
    logits, acc = evaluate()
    assert G.rel_err(logits, g['logits_init']) < 2e-4
    assert abs(acc - float(g['acc_init'])) <= 1.0 / M.N_TEST + 1e-9
    losses = []
    for k in range(M.K):
        b, r = M.inputs(data, k)
        cx.rng = InjectedRNG(G.injected_arrays_goodgan(r), cx.device)
        tr.feed(b)
        tr.train_iteration(use_graph=False)
        losses.append(tr.losses())
    torch.cuda.synchronize()
    losses, ref = np.asarray(losses), g['losses']
    assert np.abs(losses[0] - ref[0]).max() <= 5e-4 * max(1.0, np.abs(ref[0]).max())      # first iteration: identical weights
    assert np.abs(losses - ref).max() <= 0.15, np.abs(losses - ref).max(axis=1)           # then free-running (see the CIFAR-10 test)
    for net in ('good_generator', 'discriminator', 'classifier'):
        st = cx.stores[net]
        p = np.concatenate([st.get(k).reshape(-1).astype(np.float64) for k in st.names(True)])
        s1, s2 = g['checksum/' + net]
        lr = h['cla_lr'] if net == 'classifier' else h['lr']
        assert abs(p.sum() - s1) <= 4 * lr * M.K * np.sqrt(p.size) and abs((p * p).sum() - s2) <= 2e-3 * s2, net
    assert np.abs(sample() - g['sample_final']).mean() <= 0.05
    logits, acc = evaluate()
    assert np.abs(logits - g['logits_final']).mean() <= 0.1 * max(1.0, np.abs(g['logits_final']).mean())
    assert abs(acc - float(g['acc_final'])) <= 0.1
