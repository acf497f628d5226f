"""The north star's acceptance number on the HIP path (BASELINE.json: "classifier error within +-0.3 pp of the CPU reference at equal
step count"; reference Training/Train_goodGAN.py:295-351 validation loop, :428-447 _metric).

tests/golden/cifar10_long_<fixture>_<variant>.npz hold free-running 300-iteration runs of the oracle on the synthetic class-prototype
task — fixed initial weights, batches, dropout masks and noise (tests/golden/make_golden_long.py) — with the error rate on a fixed
1 000-image test split at the fixture's checkpoints, evaluated in float64 AND in several float32 variants that differ only in the order
of their sums.  Here the HIP path makes the same runs — same inputs, its own fp32 arithmetic, nothing synchronised — and is evaluated on
the same split with the same injected evaluation noise.  The oracle is not run on the GPU box.

What "equal" can mean for a free-running trajectory is MEASURED on the oracle side, not assumed and not taken from the HIP path: while
the error falls the run is chaotic (fixture 'k300', iteration 50: float64 5.7 %, the float32 variants 27 % and 65 %; Adam's early steps are
lr * sign(g), so rounding-level gradient differences become discrete weight differences), on the plateau all variants agree.  Round 2
asked whether the HIP path's 26 - 69 % at that checkpoint was such divergence or a systematic term of one of its kernels:
tests/debug/debug_long_horizon_lag.py (profiles/r03_long_horizon_lag.txt) shows both filter-gradient routings at the same rounding-level
gradient error in every variable of every one of the first 40 iterations when started from identical weights, and the float32 controls
now show the same spread without any HIP kernel involved.

Checks, all derived from the committed control runs (no hand-set window, no exemption):
  1. every checkpoint: the HIP error lies in the controls' range at that checkpoint and its two neighbours, extended on either side by
     the width of that range (for n exchangeable runs the chance of a further one falling outside shrinks fast with n; where the controls
     agree — the plateau — the extension is 0) and by the north star's 0.3 pp;
  2. the SETTLING ITERATION (first checkpoint from which on the error stays within 0.3 pp of float64's final error) lies in the controls'
     range of settling iterations extended by its own width: a lead or lag is bounded in iterations;
  3. the final checkpoint and the mean over the last third of the run are within 0.3 pp of the controls' range (extended by its width: no
     implementation can be closer to "the" reference than the reference's float32 evaluations are to each other; the width is 0 on
     'k300') — the acceptance number.
Fixture 'hard' (class blends: a genuinely ambiguous task whose error does NOT fall to zero, last 100 iterations at 2.5x the batch sizes)
makes 3. a statement about a classifier that is still imperfect; on 'k300' the plateau is 0.0 %.  Fixture 'ref' is the 'hard' task with
the last 50 iterations at the REFERENCE's batch sizes (100 / 50 / 50 / 20 / 80: the bench configuration's launch shapes, halo-tiled kernels
and all), checkpoints every 5 iterations there.
"""
import json
import os
import sys

import numpy as np
import pytest

import gpu_common as G

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
pytestmark = pytest.mark.gpu
PP = 0.003 + 1e-9                 # +-0.3 percentage points


def _state_of(tr):
    return {n: dict(p=s.p.cpu(), m=s.m.cpu(), v=s.v.cpu(), s=s.s.cpu(), step=s.step.cpu(), ema=None if s.ema is None else s.ema.cpu())
            for n, s in tr.cx.stores.items()}


def _load_state(tr, state):
    for n, s in tr.cx.stores.items():
        for k in ('p', 'm', 'v', 's', 'step'):
            getattr(s, k).copy_(state[n][k])
        if s.ema is not None:
            s.ema.copy_(state[n]['ema'])


def run_hip(M, fixture, policy=None):
    """the fixture's run on the HIP path -> ({checkpoint: error rate}, per-iteration losses)."""
    import torch
    from oracle import step_cifar10 as S
    from tg import lib
    from tg.runtime import InjectedRNG
    xt, yt, noise = M.test_split(fixture)
    steps = [e for e in M.FIXTURES[fixture]['evals']]
    total = M.total_steps(fixture)
    if total not in steps:
        steps.append(total)
    err, losses, k, state, tr = {}, [], 0, None, None
    was = lib.call('tg_conv3x3_policy', policy) if policy is not None else None
    try:
        for n_it, sizes in M.FIXTURES[fixture]['phases']:
            if tr is not None:
                state = _state_of(tr)
            tr = G.fresh_trainer(G.make_config(sizes), S.init_params(0) if state is None else None)
            if state is not None:
                _load_state(tr, state)               # the next phase's batch sizes: a new trainer (static placeholders) on the same state
            tr.set_hyper(M.HYPER['lr'], M.HYPER['cla_lr'], M.HYPER['lambda_1'], M.HYPER['lambda_2'])
            cx = tr.cx

            def error_rate():
                cx.rng = InjectedRNG({'val/C/noise': noise}, cx.device)
                return 1.0 - tr.evaluate([(xt, yt)])

            if k == 0:
                err[0] = error_rate()
            for _ in range(n_it):
                b, r = M.inputs(k, fixture)
                cx.rng = InjectedRNG(G.injected_arrays(r), cx.device)
                tr.feed(b)
                tr.train_iteration(use_graph=False)
                losses.append(tr.losses())
                k += 1
                if k in steps:
                    err[k] = error_rate()
        torch.cuda.synchronize()
    finally:
        if was is not None:
            lib.call('tg_conv3x3_policy', was)
    return err, np.asarray(losses)


def settling_step(steps, errs, level):
    """first checkpoint from which on every error is <= level."""
    s = steps[-1]
    for st, e in zip(reversed(steps), reversed(errs)):
        if e > level:
            break
        s = st
    return s


def _fixtures():
    import make_golden_long as M
    return M.committed()


@pytest.mark.parametrize("fixture", _fixtures())
def test_error_rate_stays_inside_the_envelope_of_the_cpu_reference_runs(fixture):
    import make_golden_long as M
    ctl = M.load(fixture)
    assert len(ctl) >= 3, "a float64 run and at least two float32 controls"
    steps = [int(s) for s in ctl['f64']['eval_steps']]
    for v in ctl.values():
        assert [int(s) for s in v['eval_steps']] == steps
    cerr = {name: 1.0 - v['eval_acc'] for name, v in ctl.items()}            # variant -> errors at the checkpoints
    got, losses = run_hip(M, fixture)
    herr = [got[s] for s in steps]
    table = [dict(step=s, hip_error=float(h), **{name: float(e[i]) for name, e in cerr.items()}) for i, (s, h) in enumerate(zip(steps, herr))]
    dbg = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(dbg):
        json.dump(dict(fixture=fixture, table=table, hip_losses=losses.tolist()), open(os.path.join(dbg, 'long_horizon_%s.json' % fixture), 'w'), indent=1)
    # identical weights, deterministic evaluation: the initial error is the same number (an arg-max tie at most)
    assert abs(herr[0] - cerr['f64'][0]) <= 1.0 / M.N_TEST + 1e-9, table[0]
    # 1. per checkpoint, against the controls' range in a window of one checkpoint either side
    all_err = np.stack(list(cerr.values()))                                   # [variant, checkpoint]
    for i, s in enumerate(steps):
        win = all_err[:, max(0, i - 1):i + 2]
        lo, hi = float(win.min()), float(win.max())
        w = hi - lo
        assert lo - w - PP <= herr[i] <= hi + w + PP, ('checkpoint', s, herr[i], (lo, hi), table)
    # 2. the settling iteration
    level = float(cerr['f64'][-1]) + PP
    c_settle = [settling_step(steps, list(e), level) for e in cerr.values()]
    h_settle = settling_step(steps, herr, level)
    ws = max(c_settle) - min(c_settle)
    assert min(c_settle) - ws <= h_settle <= max(c_settle) + ws, ('settling iteration', h_settle, sorted(c_settle), table)
    # 3. the acceptance number: the end of the run and the mean over its last third
    tail = [i for i, s in enumerate(steps) if s > steps[-1] * 2 // 3]
    for name, idx in (('final', [len(steps) - 1]), ('mean over the last third', tail)):
        c = [float(np.mean(e[idx])) for e in cerr.values()]
        h = float(np.mean([herr[i] for i in idx]))
        w = max(c) - min(c)                          # 0 where the reference reproduces itself ('k300'): then this IS +-0.3 pp
        assert min(c) - w - PP <= h <= max(c) + w + PP, (name, h, (min(c), max(c)), table)
    # the losses stay on the controls' scale (GAN losses fluctuate: bound = the controls' spread around float64 over the neighbouring
    # 25 iterations, per loss)
    ref = ctl['f64']['losses']
    dev = np.max([np.abs(v['losses'] - ref) for n, v in ctl.items() if n != 'f64'], axis=0)      # [iteration, 3]
    for k in range(24, len(losses), 25):
        lo, hi = max(0, k - 25), min(len(losses), k + 25)
        allowed = 2.0 * dev[lo:hi].max(axis=0) + ref[lo:hi].std(axis=0) + 0.05
        assert np.all(np.abs(losses[k] - ref[k]) <= allowed), (k + 1, losses[k], ref[k], allowed)
