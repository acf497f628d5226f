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


def test_float32_controls_of_the_golden_run_are_float32_evaluations_of_it():
    """tests/golden/cifar10_small_k10_controls.npz (make_golden.py controls): the drift budget of tests/test_gpu_golden.py — the first
    iteration of every control is the float64 one to float32 accuracy, later ones drift (that is what the file measures)."""
    import make_golden as M
    g, c = golden(), np.load(M.controls_path())
    for v in M.CONTROLS:
        assert c[v + '/losses'].shape == g['losses'].shape and c[v + '/sample_final'].shape == g['sample_final'].shape
        np.testing.assert_allclose(c[v + '/losses'][0], g['losses'][0], rtol=2e-3, atol=2e-3)
        d = float(np.abs(c[v + '/sample_final'] - g['sample_final']).mean())
        assert 1e-3 < d < 0.2, (v, d)                         # a visibly different, not a diverged trajectory
    out = M.run_control('f32a', k_steps=1)
    np.testing.assert_allclose(out['losses'][0], c['f32a/losses'][0], rtol=1e-4, atol=1e-4)


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


def test_svhn_bf16_reference_size_fixture_is_what_the_oracle_computes():
    """tests/golden/goodgan_svhn_bf16_step_ref.npz (make_golden_svhn_bf16_step.py: configs[3] at 100 / 50 / 50 / 20 / 80, float64 oracle with its
    bf16 emulation, D- / G- / C-update each from the initial weights): the cheapest of the three solver runs — the G-update, 3 s — is recomputed
    and digested here; the file holds every variable of every run and its own invariants."""
    import copy
    import make_golden_svhn_bf16_step as M
    from oracle import step_goodgan as S
    from oracle import tf_ops as T
    fx = M.load()
    P0 = M.init_params()
    b, rnd = M.inputs()
    T.MFMA_BF16 = True
    try:
        st = S.new_state(M.f64(P0))
        loss = S.g_phase(st, M.DATA, M.f64(b), M.f64(rnd)['G'], M.HYPER)
    finally:
        T.MFMA_BF16 = False
    np.testing.assert_allclose(loss, float(fx['loss/G']), rtol=1e-9)
    for k, g in st['last_grads']['G'].items():
        d = M.summary(k, g)
        np.testing.assert_allclose(d['l2'], float(fx['grad/G/%s/l2' % k]), rtol=1e-9)
        np.testing.assert_allclose(d['proj'], fx['grad/G/%s/proj' % k], rtol=1e-7, atol=1e-9 * d['l2'])
        np.testing.assert_allclose(d['sample'], fx['grad/G/%s/sample' % k], rtol=1e-6, atol=1e-7 * d['amax'])      # stored as float32
    for phase, n_vars in (('D', 0), ('G', 0), ('C', 0)):
        names = [k for k in fx if k.startswith('grad/%s/' % phase) and k.endswith('/l2')]
        assert len(names) >= 10 and all(np.isfinite(fx[k]) and float(fx[k]) > 0 for k in names), phase
    assert fx['d_labels_logits/unl'].shape == (M.SIZES['U_C'], 10) and fx['d_labels_logits/unl_d'].shape == (M.SIZES['U_D'], 10)
    assert fx['c_labels_logits/unl'].shape == (M.SIZES['U_C'], 10)
    # with bf16 operands the run is NOT the exact-fp32 one: the same G-update without the emulation differs visibly in its loss
    st2 = S.new_state(M.f64(P0))
    assert abs(S.g_phase(st2, M.DATA, M.f64(b), M.f64(rnd)['G'], M.HYPER) - loss) > 1e-6


def test_oracle_reproduces_the_long_horizon_prefix():
    """tests/golden/cifar10_long_<fixture>_<variant>.npz (make_golden_long.py: 300 free-running iterations, error rate on 1 000 images at
    the fixture's checkpoints, float64 + float32 controls) — the first iteration of every fixture is recomputed here (float64 to 1e-9; the
    float32 controls to float32 accuracy: BLAS kernels differ between hosts); the files' own invariants are checked."""
    import make_golden_long as M
    from oracle import step_cifar10 as S
    assert {'k300', 'hard'} <= set(M.committed())
    for fixture in M.committed():
        ctl = M.load(fixture)
        assert len(ctl) >= 3, (fixture, sorted(ctl))                      # float64 and at least two float32 controls
        K = M.total_steps(fixture)
        want_steps = sorted(set([0, K] + list(M.FIXTURES[fixture]['evals'])))
        for name, g in ctl.items():
            assert g['losses'].shape == (K, 3) and np.isfinite(g['losses']).all(), (fixture, name)
            assert [int(s) for s in g['eval_steps']] == want_steps and g['logits_final'].shape == (M.N_TEST, 10), (fixture, name)
            assert g['eval_acc'][0] <= 0.3                                    # starts at chance level
        err = {n: 1.0 - g['eval_acc'] for n, g in ctl.items()}
        if fixture == 'k300':
            assert all(e[-1] == 0.0 for e in err.values())                    # the saturating task: learnt by every variant
            assert max(e[10] for e in err.values()) - min(e[10] for e in err.values()) > 0.1     # ... along visibly different trajectories (iteration 50)
        else:
            assert all(0.03 <= e[-1] <= 0.45 for e in err.values()), {n: float(e[-1]) for n, e in err.items()}      # a plateau that is not 0
        zca = tuple(np.asarray(a, np.float64) for a in S.synth_zca())
        b, r = M.inputs(0, fixture)
        st = S.new_state(M.f64(S.init_params(0)))
        l64 = S.train_step(st, M.f64(b), M.f64(r), M.HYPER, zca)
        np.testing.assert_allclose(l64, ctl['f64']['losses'][0], rtol=1e-9, atol=1e-12)
        for name, g in ctl.items():
            np.testing.assert_allclose(g['losses'][0], l64, rtol=2e-3, atol=2e-3)         # one iteration: float32 rounding only
