#!/usr/bin/env python3
"""Generates tests/golden/goodgan_{mnist,svhn}_k5.npz — golden vectors for the Model/Good_GAN.py rows of SURVEY §8a (T4, T9-T12),
the same quantities as tests/golden/make_golden.py holds for Good_GAN_cifar10:

  * (d_loss, g_loss, c_loss) of K = 5 free-running iterations from fixed weights, fixed batches, fixed masks / noise;
  * per-network parameter checksums after the 5 iterations;
  * good_sampler output for fixed latents, with the initial and with the final weights;
  * classifier logits and accuracy on a fixed 100-image synthetic test split (evaluation mode), initial and final weights.

As there, the reference cannot be imported (TensorFlow 1.x absent): the vectors come from the float64 RESTATEMENT
(oracle/nets_goodgan.py, oracle/step_goodgan.py) — "parity unpinned", DESIGN.md §2.

    python tests/golden/make_golden_goodgan.py          (about a minute of NumPy float64)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import nets_goodgan as N  # noqa: E402
from oracle import step_goodgan as S  # noqa: E402

DATASETS = ('mnist', 'svhn')
SIZES = dict(B_G=6, L_C=4, U_C=4, L_D=2, U_D=4)
# Training/Train_goodGAN.py:673,680 (mnist: G/D 1e-3, C 3e-4), :505 (svhn: 3e-4 everywhere); lambda_1 as late in the schedule
HYPER = {'mnist': dict(lr=1e-3, cla_lr=3e-4, beta1=0.5, lambda_1=0.1, lambda_2=0.0),
         'svhn': dict(lr=3e-4, cla_lr=3e-4, beta1=0.5, lambda_1=0.1, lambda_2=0.0)}
K = 5
N_TEST = 100
N_SAMPLE = 6


def f64(d):
    return {k: (f64(v) if isinstance(v, dict) else np.asarray(v, np.float64)) for k, v in d.items()}


def init_params(data):
    """float32 initial weights: the reference's shapes with benign values (its mean-.02 / stddev-1 initialisers saturate every
    unit at these sizes — SURVEY T18), moving statistics and affine terms away from their trivial values."""
    P = N.init_params(data, 11, np.float64)
    rng = np.random.default_rng(12)
    for k in P:
        if k.endswith('kernel'):
            P[k] = P[k] * 0.05
        elif k.endswith(('/g', 'gamma')):
            P[k] = 1 + 0.3 * rng.standard_normal(P[k].shape)
        elif k.endswith(('/b', 'bias', 'beta')):
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
        elif k.endswith('moving_variance'):
            P[k] = 0.5 + rng.random(P[k].shape)
        elif k.endswith('moving_mean'):
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    return {k: v.astype(np.float32) for k, v in P.items()}


def inputs(data, k):
    return S.synth_batch(data, 300 + k, SIZES), S.synth_rnd(data, 400 + k, SIZES)


def sample_latents():
    z = np.random.default_rng(7).uniform(-1, 1, (N_SAMPLE, 100)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.arange(N_SAMPLE) % 10]
    return z, y


def test_split(data):
    sizes = dict(SIZES, L_C=N_TEST)
    b = S.synth_batch(data, 999, sizes)
    rnd = S.synth_rnd(data, 998, sizes)['C']['C_real']          # evaluation mode uses the input noise only (dropout is off)
    return b['x_l_c'], b['y_l_c'], rnd


def sample(P, data):
    z, y = sample_latents()
    return N.seq_fwd(P, N.generator_layers(data), z.astype(np.float64), y.astype(np.float64), {}, True)[0]


def evaluate(P, data):
    x, y, rnd = test_split(data)
    logits = N.seq_fwd(P, N.classifier_layers(data), x.astype(np.float64), None, f64(rnd), False)[0]
    return logits, float((logits.argmax(1) == y.argmax(1)).mean())


def checksums(P):
    out = {}
    for net in ('good_generator', 'discriminator', 'classifier'):
        vals = [v for k, v in P.items() if k.startswith(net + '/') and 'pop_mean' not in k and 'moving_' not in k]
        out[net] = np.array([sum(float(v.sum()) for v in vals), sum(float((v * v).sum()) for v in vals)])
    return out


def run(data, k_steps=K):
    st = S.new_state(f64(init_params(data)))
    out = {'sample_init': sample(st['P'], data)}
    out['logits_init'], out['acc_init'] = evaluate(st['P'], data)
    losses = []
    for k in range(k_steps):
        b, r = inputs(data, k)
        losses.append(S.train_step(st, data, f64(b), f64(r), HYPER[data]))
    out['losses'] = np.asarray(losses)
    out['sample_final'] = sample(st['P'], data)
    out['logits_final'], out['acc_final'] = evaluate(st['P'], data)
    for net, v in checksums(st['P']).items():
        out['checksum/' + net] = v
    return out


def path(data):
    return os.path.join(HERE, 'goodgan_%s_k%d.npz' % (data, K))


if __name__ == "__main__":
    for data in DATASETS:
        g = run(data)
        np.savez_compressed(path(data), **{k: (np.asarray(v, np.float32) if k.startswith('sample') else np.asarray(v)) for k, v in g.items()})
        print('wrote', path(data), os.path.getsize(path(data)), 'bytes; losses[0]', g['losses'][0], 'acc', g['acc_init'], g['acc_final'])
