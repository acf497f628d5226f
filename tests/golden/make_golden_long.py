#!/usr/bin/env python3
"""Generates tests/golden/cifar10_long_k<K>.npz — the north star's acceptance number (BASELINE.json: "classifier error within
+-0.3 pp of the CPU reference at equal step count"; reference Training/Train_goodGAN.py:295-351 validation loop, :428-447 _metric)
as a golden vector: the float64 RESTATEMENT (oracle/; the reference itself cannot run here, SURVEY §8c — "parity unpinned") trains the
CIFAR-10 model free-running for K small-batch iterations on the synthetic class-prototype task of SURVEY §8d from fixed initial
weights, fixed batches and fixed masks / noise, and its error rate on a fixed 1 000-image test split is recorded every EVAL_EVERY
iterations (evaluation mode: pop_mean, no dropout, the always-on input noise injected from a fixed seed).

tests/test_gpu_long_horizon.py runs the HIP path on the same inputs (without running the oracle on the GPU box) and compares the
error rates; tests/test_golden.py re-checks the first iterations of this file against the oracle on the CPU.

    python tests/golden/make_golden_long.py [K]          (about 4 s of NumPy float64 per iteration on 8 cores)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nets_cifar10 as N  # noqa: E402
from oracle import step_cifar10 as S  # noqa: E402

SIZES = dict(B_G=10, L_C=10, U_C=10, L_D=4, U_D=6)
HYPER = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)
K = 300
EVAL_EVERY = 25
N_TEST = 1000
NOISE = 0.25            # the synthetic task's pixel noise (S.synth_batch)


def f64(d):
    return {k: (f64(v) if isinstance(v, dict) else np.asarray(v, np.float64)) for k, v in d.items()}


def inputs(k):
    full = dict(S.SIZES, **SIZES)
    return S.synth_batch(1000 + k, full), S.synth_rnd(5000 + k, full)


def test_split():
    b = S.synth_batch(99999, dict(S.SIZES, L_C=N_TEST))
    noise = (0.15 * np.random.default_rng(99998).standard_normal(b['x_l_c'].shape)).astype(np.float32)
    return b['x_l_c'], b['y_l_c'], noise


def evaluate(P, zca, split):
    x, y, noise = split
    correct = 0
    logits = []
    for i in range(0, N_TEST, 250):
        lg, _, _ = N.classifier_fwd(P, N.zca_apply(x[i:i + 250].astype(np.float64), *zca), False, {'noise': noise[i:i + 250].astype(np.float64)})
        logits.append(lg)
        correct += int((lg.argmax(1) == y[i:i + 250].argmax(1)).sum())
    return np.concatenate(logits), correct / float(N_TEST)


def path(k=K):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cifar10_long_k%d.npz' % k)


def run(k_steps=K, log=None):
    st = S.new_state(f64(S.init_params(0)))
    zca = tuple(np.asarray(a, np.float64) for a in S.synth_zca())
    split = test_split()
    losses, evals = [], []
    _, acc = evaluate(st['P'], zca, split)
    evals.append((0, acc))
    t0 = time.time()
    for k in range(k_steps):
        b, r = inputs(k)
        losses.append(S.train_step(st, f64(b), f64(r), HYPER, zca))
        if (k + 1) % EVAL_EVERY == 0 or k + 1 == k_steps:
            logits, acc = evaluate(st['P'], zca, split)
            evals.append((k + 1, acc))
            if log:
                log("step %d  losses %s  test error %.4f  (%.0f s)" % (k + 1, np.round(losses[-1], 4), 1 - acc, time.time() - t0))
    return dict(losses=np.asarray(losses), eval_steps=np.asarray([e[0] for e in evals]), eval_acc=np.asarray([e[1] for e in evals]),
                logits_final=logits.astype(np.float32))


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else K
    g = run(k, log=lambda s: print(s, flush=True))
    np.savez_compressed(path(k), **g)
    print('wrote', path(k), os.path.getsize(path(k)), 'bytes; error curve', list(zip(g['eval_steps'], np.round(1 - g['eval_acc'], 4))))
