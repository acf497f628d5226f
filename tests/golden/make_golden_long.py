#!/usr/bin/env python3
"""Generates the long-horizon fixtures tests/golden/cifar10_long_<fixture>_<variant>.npz — the north star's acceptance number
(BASELINE.json: "classifier error within +-0.3 pp of the CPU reference at equal step count"; reference
Training/Train_goodGAN.py:295-351 validation loop, :428-447 _metric) as golden vectors: the RESTATEMENT (oracle/; the reference itself
cannot run here, SURVEY §8c — "parity unpinned") trains the CIFAR-10 model free-running for K iterations on the synthetic
class-prototype task of SURVEY §8d from fixed initial weights, fixed batches and fixed masks / noise, and its error rate on a fixed
1 000-image test split is recorded at the fixture's checkpoints (evaluation mode: pop_mean, no dropout, the always-on input noise
injected from a fixed seed).

Fixtures (FIXTURES):
  * 'k300'  — round 2's task (pixel noise 0.25, batches 10/10/10/4/6): the error falls from 90 % to 0 within 75 iterations and stays
              there.  Checkpoints every 5 iterations through the transient, so that a lead or lag is measured in ITERATIONS.
  * 'hard'  — a task whose error does NOT fall to zero: every image is a blend a*proto[y] + (1-a)*proto[y'] with a ~ U(0.5, 1) of its own
              class and a random other one (images near a = 0.5 are genuinely ambiguous) under pixel noise 0.75; the error settles at
              10 - 15 % (tuning runs: 13.5 / 14.3 / 12.3 / 11.5 % at iterations 225 ... 300 with pixel noise 0.75, 7 - 8.5 % with 0.4); the
              last 100 iterations run at 2.5x the batch sizes: +-0.3 pp is checked where a classifier that merely "works" does not pass.
  * 'ref'   — the 'hard' task with the last 50 iterations at the REFERENCE's batch sizes (100 / 50 / 50 / 20 / 80, the bench configuration: 38 s
              per float64 iteration of the oracle on 8 cores, which is why only 50), checkpoints every 5 iterations there.

Variants per fixture (VARIANTS) — the SAME run evaluated several ways:
  * 'f64'          — float64 (the golden trajectory),
  * 'f32a' ... 'f32d' — float32 with the reduction order of every conv / dense contraction and of the filter-gradient sums varied: as
                     NumPy / BLAS has it or BACKWARDS (oracle.tf_ops.SUM_REVERSED), the im2col batch chunks as they are or 8x smaller
                     (another partition of the filter-gradient accumulation; a 3x smaller chunk changes nothing at these batch sizes).
The float32 variants are correct float32 evaluations of the reference's arithmetic that differ ONLY in rounding.  Their distance from f64
and from each other is what a correct float32 implementation can be expected to show on this free-running trajectory (it is large while
the error falls: the run is chaotic there — at iteration 50 of 'k300' they sit at 65 % and 27 % against float64's 6 %); the HIP path is
required to stay inside the envelope of the committed variants (tests/test_gpu_long_horizon.py), and no bound in that test is set by hand
or taken from the HIP path's own behaviour.  Not every variant needs to exist for every fixture: load() returns those that are committed.

tests/test_golden.py re-checks the first iterations of every committed file against the oracle on the CPU.

    python tests/golden/make_golden_long.py <fixture> <variant> [K]
    (k300: ~25 min float32 / ~45 min float64 on 8 cores; hard: ~1.5x that)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nets_cifar10 as N  # noqa: E402
from oracle import step_cifar10 as S  # noqa: E402
from oracle import tf_ops as T  # noqa: E402

SIZES = dict(B_G=10, L_C=10, U_C=10, L_D=4, U_D=6)
SIZES_LATE = dict(B_G=25, L_C=25, U_C=25, L_D=10, U_D=15)
SIZES_REF = dict(B_G=100, L_C=50, U_C=50, L_D=20, U_D=80)     # the reference's own batch sizes (Training/Train_goodGAN.py:566-572): BASELINE configs[1]
HYPER = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)
N_TEST = 1000
EVAL_CHUNK = 25         # images per evaluation pass (small arrays stay inside the allocator's heap: 4x faster than 250 here)

FIXTURES = {
    # phases: (number of iterations, batch sizes); evals: checkpoints (iteration counts, 0 is always evaluated)
    'k300': dict(noise=0.25, phases=((300, SIZES),),
                 evals=sorted(set(range(5, 101, 5)) | set(range(25, 301, 25)))),
    'hard': dict(noise=0.75, mix=0.5, phases=((200, SIZES), (100, SIZES_LATE)),
                 evals=sorted(set(range(25, 301, 25)) | set(range(210, 301, 10)))),
    'ref': dict(noise=0.75, mix=0.5, phases=((250, SIZES), (50, SIZES_REF)),
                 evals=sorted(set(range(25, 301, 25)) | set(range(255, 301, 5)))),
}
VARIANTS = {
    'f64': dict(dtype=np.float64, reversed=False, chunk=1),
    'f32a': dict(dtype=np.float32, reversed=False, chunk=1),
    'f32b': dict(dtype=np.float32, reversed=True, chunk=8),
    'f32c': dict(dtype=np.float32, reversed=True, chunk=1),
    'f32d': dict(dtype=np.float32, reversed=False, chunk=8),
}
K = 300                 # every fixture's length


def cast(d, dtype):
    return {k: (cast(v, dtype) if isinstance(v, dict) else np.asarray(v, dtype)) for k, v in d.items()}


def f64(d):
    return cast(d, np.float64)


def sizes_at(fixture, k):
    """batch sizes of iteration k (0-based) of a fixture."""
    for n, sizes in FIXTURES[fixture]['phases']:
        if k < n:
            return sizes
        k -= n
    raise IndexError(k)


def total_steps(fixture):
    return sum(n for n, _ in FIXTURES[fixture]['phases'])


def inputs(k, fixture='k300'):
    full = dict(S.SIZES, **sizes_at(fixture, k))
    return S.synth_batch(1000 + k, full, noise=FIXTURES[fixture]['noise'], mix=FIXTURES[fixture].get('mix', 0.0)), S.synth_rnd(5000 + k, full)


def test_split(fixture='k300'):
    b = S.synth_batch(99999, dict(S.SIZES, L_C=N_TEST), noise=FIXTURES[fixture]['noise'], mix=FIXTURES[fixture].get('mix', 0.0))
    noise = (0.15 * np.random.default_rng(99998).standard_normal(b['x_l_c'].shape)).astype(np.float32)
    return b['x_l_c'], b['y_l_c'], noise


def evaluate(P, zca, split, n_test=N_TEST):
    x, y, noise = split
    dt = zca[1].dtype
    correct = 0
    logits = []
    for i in range(0, n_test, EVAL_CHUNK):
        lg, _, _ = N.classifier_fwd(P, N.zca_apply(x[i:i + EVAL_CHUNK].astype(dt), *zca), False, {'noise': noise[i:i + EVAL_CHUNK].astype(dt)})
        logits.append(lg)
        correct += int((lg.argmax(1) == y[i:i + EVAL_CHUNK].argmax(1)).sum())
    return np.concatenate(logits), correct / float(n_test)


def path(fixture='k300', variant='f64'):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cifar10_long_%s_%s.npz' % (fixture, variant))


def run(fixture='k300', variant='f64', k_steps=None, log=None, n_test=N_TEST, evals=None):
    v = VARIANTS[variant]
    dt = v['dtype']
    k_steps = total_steps(fixture) if k_steps is None else k_steps
    evals = [e for e in (FIXTURES[fixture]['evals'] if evals is None else evals) if e <= k_steps]
    if k_steps not in evals:
        evals.append(k_steps)
    saved = T.SUM_REVERSED, T._CHUNK_ELEMS
    T.SUM_REVERSED, T._CHUNK_ELEMS = v['reversed'], T._CHUNK_ELEMS // v['chunk']
    try:
        st = S.new_state(cast(S.init_params(0), dt))
        zca = tuple(np.asarray(a, dt) for a in S.synth_zca())
        split = test_split(fixture)
        losses, table = [], []
        logits, acc = evaluate(st['P'], zca, split, n_test)
        table.append((0, acc))
        t0 = time.time()
        for k in range(k_steps):
            b, r = inputs(k, fixture)
            losses.append(S.train_step(st, cast(b, dt), cast(r, dt), HYPER, zca))
            if k + 1 in evals:
                logits, acc = evaluate(st['P'], zca, split, n_test)
                table.append((k + 1, acc))
                if log:
                    log("%s/%s step %d  losses %s  test error %.4f  (%.0f s)" % (fixture, variant, k + 1, np.round(losses[-1], 4), 1 - acc, time.time() - t0))
    finally:
        T.SUM_REVERSED, T._CHUNK_ELEMS = saved
    return dict(losses=np.asarray(losses, np.float64), eval_steps=np.asarray([e[0] for e in table]),
                eval_acc=np.asarray([e[1] for e in table]), logits_final=logits.astype(np.float32))


def committed():
    """the fixtures whose float64 trajectory is in the tree (a fixture is generated variant by variant; the tests take those that are there)."""
    return [f for f in FIXTURES if os.path.exists(path(f, 'f64'))]


def load(fixture):
    """{variant: npz} of the committed files of one fixture ('f64' must be among them)."""
    out = {v: np.load(path(fixture, v)) for v in VARIANTS if os.path.exists(path(fixture, v))}
    assert 'f64' in out, 'the float64 trajectory of %r is missing' % fixture
    return out


if __name__ == "__main__":
    fixture, variant = sys.argv[1], sys.argv[2]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else None
    g = run(fixture, variant, k, log=lambda s: print(s, flush=True))
    out = path(fixture, variant) if k is None else '/tmp/cifar10_long_%s_%s_k%d.npz' % (fixture, variant, k)
    np.savez_compressed(out, **g)
    print('wrote', out, os.path.getsize(out), 'bytes; error curve', list(zip(g['eval_steps'].tolist(), np.round(1 - g['eval_acc'], 4).tolist())))
