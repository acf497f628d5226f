"""Golden vectors of SURVEY §8c (tests/golden/cifar10_small_k10.npz, written by tests/golden/make_golden.py from the float64
restatement; goodgan_{mnist,svhn}_k5.npz by make_golden_goodgan.py): the oracle must keep reproducing them (guards the checker against accidental edits)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))


def golden():
    return np.load(os.path.join(HERE, 'golden', 'cifar10_small_k10.npz'))


def test_oracle_reproduces_the_golden_prefix():
    import make_golden as M
    g = golden()
    out = M.run(k_steps=2)                                   # two of the ten iterations keep the CPU suite short
    np.testing.assert_allclose(out['losses'], g['losses'][:2], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out['sample_init'], g['sample_init'], rtol=0, atol=1e-6)        # stored as float32
    np.testing.assert_allclose(out['logits_init'], g['logits_init'], rtol=1e-9, atol=1e-12)
    assert out['acc_init'] == float(g['acc_init'])
    assert g['losses'].shape == (10, 3) and g['sample_final'].shape == (8, 32, 32, 3) and g['logits_final'].shape == (M.N_TEST, 10)


def test_goodgan_oracle_reproduces_the_golden_prefix():
    import make_golden_goodgan as M
    for data in M.DATASETS:
        g = np.load(M.path(data))
        out = M.run(data, k_steps=1)                         # one of the five iterations keeps the CPU suite short
        np.testing.assert_allclose(out['losses'], g['losses'][:1], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(out['sample_init'], g['sample_init'], rtol=0, atol=1e-6)    # stored as float32
        np.testing.assert_allclose(out['logits_init'], g['logits_init'], rtol=1e-9, atol=1e-12)
        assert out['acc_init'] == float(g['acc_init'])
        assert g['losses'].shape == (M.K, 3) and g['logits_final'].shape == (M.N_TEST, 10)
        assert g['sample_final'].shape[0] == M.N_SAMPLE and np.isfinite(g['losses']).all()


def test_oracle_reproduces_the_long_horizon_prefix():
    """tests/golden/cifar10_long_k300.npz (make_golden_long.py: 300 free-running iterations, error rate on 1 000 images every 25) —
    the first iteration and the initial error rate are recomputed here; the file's own invariants are checked."""
    import make_golden_long as M
    from oracle import step_cifar10 as S
    g = np.load(M.path(M.K))
    assert g['losses'].shape == (M.K, 3) and np.isfinite(g['losses']).all()
    assert list(g['eval_steps']) == [0] + list(range(M.EVAL_EVERY, M.K + 1, M.EVAL_EVERY)) and g['logits_final'].shape == (M.N_TEST, 10)
    assert g['eval_acc'][-1] >= 0.99 and g['eval_acc'][0] <= 0.3               # the task is learnt within the run
    st = S.new_state(M.f64(S.init_params(0)))
    zca = tuple(np.asarray(a, np.float64) for a in S.synth_zca())
    b, r = M.inputs(0)
    np.testing.assert_allclose(S.train_step(st, M.f64(b), M.f64(r), M.HYPER, zca), g['losses'][0], rtol=1e-9, atol=1e-12)
    x, y, noise = M.test_split()
    P0 = M.f64(S.init_params(0))
    from oracle import nets_cifar10 as N
    lg, _, _ = N.classifier_fwd(P0, N.zca_apply(x[:100].astype(np.float64), *zca), False, {'noise': noise[:100].astype(np.float64)})
    assert lg.shape == (100, 10)
