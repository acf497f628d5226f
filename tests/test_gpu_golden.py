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

# CIFAR-10 file (round 3): the free-running bounds — losses per iteration, sampler output, evaluation logits — are the measured spread of
# the oracle's own float32 evaluations of the same run (cifar10_small_k10_controls.npz), extended by its width; see the test body.
# MNIST / SVHN files: free-running bounds (no synchronisation with the oracle), derived as in tests/test_gpu_step.py::test_free_running_three_iterations:
#   * losses: 1.5e-2 * max(1, |ref|) per elapsed iteration — the size of the drift two equally accurate implementations show once
#     sign-like early Adam steps have put rounding-noise elements 2*lr apart;
#   * evaluation after K iterations: the logits may differ by the accumulated drift, but an image can only change its arg-max where the
#     golden margin (top-1 minus top-2 logit) is smaller than twice the logit deviation of that image — so the accuracy difference is
#     bounded by the NUMBER of such images, counted on the committed golden logits, not by a blanket percentage.  (Measured, round 2:
#     MNIST 0 fragile images, SVHN 1 %, and ALL of them for the CIFAR-10 file: ten 4-image iterations at classifier lr 3e-3 leave the
#     evaluation logits at chance level — |logit| ~ 0.28, accuracy 10.5 % — so that file's final accuracy carries no information;
#     the error rate of a run that learns is pinned by tests/test_gpu_long_horizon.py.)
#   * the logits themselves: their deviation must stay below the drift budget or, where the golden logits are still noise-sized,
#     below the golden logits' own mean magnitude.
LOSS_DRIFT = 1.5e-2


def _loss_envelope(losses, ref):
    k = np.arange(1, len(ref) + 1)[:, None]
    return np.abs(losses - ref) <= LOSS_DRIFT * np.maximum(1.0, np.abs(ref)) * k


def _accuracy_bound(logits, golden_logits, labels):
    """(|accuracy difference| allowed, measured) from the per-image margin argument above."""
    gl = np.asarray(golden_logits, np.float64)
    dev = np.abs(np.asarray(logits, np.float64) - gl).max(axis=1)
    top2 = np.sort(gl, axis=1)[:, -2:]
    fragile = (top2[:, 1] - top2[:, 0]) <= 2.0 * dev
    same = logits.argmax(1) == gl.argmax(1)
    assert same[~fragile].all()                                 # outside the fragile set the decision cannot differ
    acc, acc_g = (logits.argmax(1) == labels.argmax(1)).mean(), (gl.argmax(1) == labels.argmax(1)).mean()
    return fragile.sum() / float(len(gl)), abs(acc - acc_g), dev


def _dump(name, **kw):
    dbg = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(dbg):
        import json
        json.dump({k: (np.asarray(v).tolist()) for k, v in kw.items()}, open(os.path.join(dbg, name), 'w'))


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
        with cx.phase_scope('val', record=False):
            with cx.rng_scoped('val/C'):
                logits, _ = tr.model.classifier(tr.model.zca().apply(cx.from_numpy(xt)), False)
        cx.rng = InjectedRNG({'val/C/noise': noise}, cx.device)
        return logits.numpy(), tr.evaluate([(xt, yt)])

    # initial weights: sampler and evaluation are deterministic functions of the inputs -> fp32 tolerance
    assert G.rel_err(tr.sample(z, y), g['sample_init']) < 2e-4
    logits0, acc0 = evaluate()
    assert G.rel_err(logits0, g['logits_init']) < 2e-4
    assert abs(acc0 - float(g['acc_init'])) <= 1.0 / M.N_TEST + 1e-9            # one arg-max tie at most
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
    # afterwards the trajectories drift (sign-like first Adam steps, tests/test_gpu_step.py docstring).  How far is measured on the oracle
    # side: tests/golden/cifar10_small_k10_controls.npz holds the oracle's float32 evaluations of this run with other summation orders
    # (make_golden.py controls).  After ten iterations any two of the five reference trajectories {float64, four distinct float32} are
    # 0.12 - 0.27 apart in the evaluation logits (float64 is not closer to the others than they are to each other) — the run is chaotic
    # at this size.  Yardstick for every free-running quantity: the PAIRWISE distances among the reference trajectories; the HIP path's
    # distance to the float64 file must not exceed their maximum extended by their spread.  Per iteration k for the losses (running
    # maximum: what the references have reached by then).
    ctl = np.load(M.controls_path())
    refs = [ref] + [ctl[v + '/losses'] for v in M.CONTROLS]
    pair = np.stack([np.abs(a - b) for i, a in enumerate(refs) for b in refs[i + 1:]])          # [pair, iteration, loss]
    nz = pair.reshape(len(pair), -1).max(axis=1) > 0                                             # (two variants coincide at this batch size)
    env = np.maximum.accumulate(2 * pair[nz].max(axis=0) - pair[nz].min(axis=0), axis=0) + 2e-4 * np.abs(ref)
    assert (np.abs(losses - ref) <= env).all(), (np.abs(losses - ref), env)
    for net in ('good_generator', 'discriminator', 'classifier'):
        st = cx.stores[net]
        p = np.concatenate([st.get(k).reshape(-1).astype(np.float64) for k in st.names(True)])
        s1, s2 = g['checksum/' + net]
        lr = M.HYPER['cla_lr'] if net == 'classifier' else M.HYPER['lr']
        # free-running trajectories: elements whose gradients are rounding noise take +-lr steps of either sign, so the plain sum
        # random-walks by ~lr*K*sqrt(N) (allowed: 4x) between two correct implementations (a wrong step size or a missing update moves it by
        # ~lr*K*N, a thousand times more); the sum of squares is insensitive to that
        assert abs(p.sum() - s1) <= 4 * lr * M.K * np.sqrt(p.size) and abs((p * p).sum() - s2) <= 2e-3 * s2, net
    smp = tr.sample(z, y)
    logits, acc = evaluate()
    allowed, measured, dev = _accuracy_bound(logits, g['logits_final'], yt)
    _dump('golden_cifar10.json', loss_dev=np.abs(losses - ref) / np.maximum(1.0, np.abs(ref)), sample_mean_abs=np.abs(smp - g['sample_final']).mean(),
          logit_dev_mean=dev.mean(), logit_dev_max=dev.max(), logit_scale=np.abs(g['logits_final']).mean(), acc=acc, acc_golden=float(g['acc_final']),
          acc_allowed=allowed, acc_measured=measured)
    # sampler output and evaluation logits after the ten iterations: the same pairwise yardstick (measured: sampler 0.020 - 0.050 mean absolute
    # difference between two reference trajectories, logits 0.12 - 0.27 mean over the images of the largest logit difference; round 2 used a
    # hand-derived 0.0375 for the sampler, which most pairs of references do not meet)
    def extended_max(dist, items):
        d = [dist(a, b) for i, a in enumerate(items) for b in items[i + 1:]]
        d = [v for v in d if v > 0]
        return 2 * max(d) - min(d)
    smps = [g['sample_final'].astype(np.float64)] + [ctl[v + '/sample_final'].astype(np.float64) for v in M.CONTROLS]
    lgts = [g['logits_final'].astype(np.float64)] + [ctl[v + '/logits_final'].astype(np.float64) for v in M.CONTROLS]
    b_smp = extended_max(lambda a, b: float(np.abs(a - b).mean()), smps)
    b_log = extended_max(lambda a, b: float(np.abs(a - b).max(axis=1).mean()), lgts)
    assert np.abs(smp - g['sample_final']).mean() <= b_smp, (np.abs(smp - g['sample_final']).mean(), b_smp)
    assert dev.mean() <= b_log, (dev.mean(), b_log)
    assert measured <= allowed + 1e-9 and abs(acc - float(g['acc_final'])) <= allowed + 1e-9, (acc, float(g['acc_final']), allowed)


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
    assert _loss_envelope(losses, ref).all(), np.abs(losses - ref) / np.maximum(1.0, np.abs(ref))   # then free-running (module header)
    for net in ('good_generator', 'discriminator', 'classifier'):
        st = cx.stores[net]
        p = np.concatenate([st.get(k).reshape(-1).astype(np.float64) for k in st.names(True)])
        s1, s2 = g['checksum/' + net]
        lr = h['cla_lr'] if net == 'classifier' else h['lr']
        assert abs(p.sum() - s1) <= 4 * lr * M.K * np.sqrt(p.size) and abs((p * p).sum() - s2) <= 2e-3 * s2, net
    smp = sample()
    logits, acc = evaluate()
    allowed, measured, dev = _accuracy_bound(logits, g['logits_final'], yt)
    _dump('golden_%s.json' % data, loss_dev=np.abs(losses - ref) / np.maximum(1.0, np.abs(ref)), sample_mean_abs=np.abs(smp - g['sample_final']).mean(),
          logit_dev_mean=dev.mean(), logit_dev_max=dev.max(), logit_scale=np.abs(g['logits_final']).mean(), acc=acc, acc_golden=float(g['acc_final']),
          acc_allowed=allowed, acc_measured=measured)
    assert np.abs(smp - g['sample_final']).mean() <= LOSS_DRIFT * M.K / 4
    assert dev.mean() <= LOSS_DRIFT * M.K * max(1.0, np.abs(g['logits_final']).mean())                # measured: 0.0015 (MNIST), 0.074 (SVHN)
    assert measured <= allowed + 1e-9 and abs(acc - float(g['acc_final'])) <= allowed + 1e-9, (acc, float(g['acc_final']), allowed)
