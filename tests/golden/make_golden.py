#!/usr/bin/env python3
"""Generates tests/golden/cifar10_small_k10.npz — the golden vectors SURVEY §8c lists for the Python harness rows (T1, T2, T25).

The reference itself cannot be imported here (TensorFlow 1.x absent), so — as the survey prescribes — the vectors are
outputs of the float64 RESTATEMENT (oracle/) on seeded inputs, i.e. they pin the restatement and the HIP path against each
other and against accidental edits, not against TensorFlow ("parity unpinned"):

  * (d_loss, g_loss, c_loss) of K = 10 free-running iterations from fixed initial weights, fixed batches, fixed masks / noise;
  * per-network parameter checksums after the 10 iterations;
  * good_sampler output for a fixed sample_z / sample_y, with the initial and with the final weights;
  * classifier logits and accuracy on a fixed 200-image synthetic test split (evaluation mode), initial and final weights.

    python tests/golden/make_golden.py          (about one minute of NumPy float64)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nets_cifar10 as N  # noqa: E402
from oracle import step_cifar10 as S  # noqa: E402

SIZES = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)
HYPER = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)
K = 10
N_TEST = 200


def f64(d):
    return {k: (f64(v) if isinstance(v, dict) else np.asarray(v, np.float64)) for k, v in d.items()}


def inputs(k):
    full = dict(S.SIZES, **SIZES)
    return S.synth_batch(100 + k, full), S.synth_rnd(200 + k, full)


def sample_latents():
    z = np.random.default_rng(7).uniform(-1, 1, (8, 100)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.arange(8) % 10]
    return z, y


def test_split():
    b = S.synth_batch(999, dict(S.SIZES, L_C=N_TEST))
    noise = (0.15 * np.random.default_rng(998).standard_normal(b['x_l_c'].shape)).astype(np.float32)
    return b['x_l_c'], b['y_l_c'], noise


def evaluate(P, zca):
    x, y, noise = test_split()
    logits, _, _ = N.classifier_fwd(P, N.zca_apply(x.astype(np.float64), *zca), False, {'noise': noise.astype(np.float64)})
    return logits, float((logits.argmax(1) == y.argmax(1)).mean())


def checksums(P):
    out = {}
    for net in ('good_generator', 'discriminator', 'classifier'):
        vals = [v for k, v in P.items() if k.startswith(net + '/') and 'pop_mean' not in k and 'moving_' not in k]   # trainable variables
        out[net] = np.array([sum(float(v.sum()) for v in vals), sum(float((v * v).sum()) for v in vals)])
    return out


def run_control(variant, k_steps=K):
    """the same run evaluated in float32 with the summation order of make_golden_long.VARIANTS[variant]: how far two correct float32
    evaluations of this free-running trajectory end up from the float64 one (the drift budget of tests/test_gpu_golden.py)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden_long as ML
    from oracle import tf_ops as T
    v = ML.VARIANTS[variant]
    dt = v['dtype']
    saved = T.SUM_REVERSED, T._CHUNK_ELEMS
    T.SUM_REVERSED, T._CHUNK_ELEMS = v['reversed'], T._CHUNK_ELEMS // v['chunk']
    try:
        st = S.new_state(ML.cast(S.init_params(0), dt))
        zca = tuple(np.asarray(a, dt) for a in S.synth_zca())
        z, y = sample_latents()
        losses = []
        for k in range(k_steps):
            b, r = inputs(k)
            losses.append(S.train_step(st, ML.cast(b, dt), ML.cast(r, dt), HYPER, zca))
        x, yt, noise = test_split()
        logits, _, _ = N.classifier_fwd(st['P'], N.zca_apply(x.astype(dt), *zca), False, {'noise': noise.astype(dt)})
        return dict(losses=np.asarray(losses, np.float64), sample_final=N.generator_fwd(st['P'], z.astype(dt), y.astype(dt))[0].astype(np.float32),
                    logits_final=np.asarray(logits, np.float32))
    finally:
        T.SUM_REVERSED, T._CHUNK_ELEMS = saved


CONTROLS = ('f32a', 'f32b', 'f32c', 'f32d')


def controls_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cifar10_small_k10_controls.npz')


def run(k_steps=K):
    P32 = S.init_params(0)
    st = S.new_state(f64(P32))
    zca = tuple(np.asarray(a, np.float64) for a in S.synth_zca())
    z, y = sample_latents()
    out = {'sample_init': N.generator_fwd(st['P'], z.astype(np.float64), y.astype(np.float64))[0]}
    out['logits_init'], out['acc_init'] = evaluate(st['P'], zca)
    losses = []
    for k in range(k_steps):
        b, r = inputs(k)
        losses.append(S.train_step(st, f64(b), f64(r), HYPER, zca))
    out['losses'] = np.asarray(losses)
    out['sample_final'] = N.generator_fwd(st['P'], z.astype(np.float64), y.astype(np.float64))[0]
    out['logits_final'], out['acc_final'] = evaluate(st['P'], zca)
    for net, v in checksums(st['P']).items():
        out['checksum/' + net] = v
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == 'controls':
        out = {}
        for v in CONTROLS:
            for k, a in run_control(v).items():
                out['%s/%s' % (v, k)] = a
        np.savez_compressed(controls_path(), **out)
        print('wrote', controls_path(), os.path.getsize(controls_path()), 'bytes')
        sys.exit(0)
    g = run()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cifar10_small_k10.npz')
    np.savez_compressed(path, **{k: (np.asarray(v, np.float32) if k.startswith('sample') else np.asarray(v)) for k, v in g.items()})
    print('wrote', path, os.path.getsize(path), 'bytes; losses[0]', g['losses'][0], 'acc', g['acc_init'], g['acc_final'])
