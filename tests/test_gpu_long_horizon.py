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

Round 4: an ENSEMBLE of HIP trajectories against the ensemble of control runs.  One HIP draw against a band three control-ranges wide
(round 3) cannot tell chance from a small systematic term; the path has several equally accurate float32 summation orders of its own —
the 3x3 layers on the halo-tiled kernels (default routing) or wherever they apply (tg_conv3x3_policy 1), the filter-gradient pixel
split as the library sets it or halved — and both launch paths (one stream / second-stream overlap; bit-identical arithmetic, so that
the overlap path is held to the acceptance number too).  MEMBERS below = 4 trajectories per fixture.

Checks, all derived from the committed control runs (no hand-set window, no exemption):
  1. every member, every checkpoint FROM THE ITERATION ON BY WHICH CHECK 2 REQUIRES A RUN TO HAVE SETTLED: inside the controls' range at that
     checkpoint and its two neighbours, extended on either side by the width of that range + 0.3 pp — on the plateau the controls agree and
     this is the north star's 0.3 pp.  Before that iteration nothing per-checkpoint is asserted: the controls themselves are 5.7 % ... 65 %
     apart at iteration 50 of 'k300' (the run is chaotic while the error falls; round 3's per-checkpoint envelope there was [-54 %, 124 %],
     i.e. vacuous, and a 5 pp clamp on it rejects late-settling HIP members the way it would reject late-settling controls) — the
     descent is judged by check 2, the full tables go to gpurun_out/long_horizon_<fixture>.json.
  2. every member: the SETTLING ITERATION (first checkpoint from which on the error stays within 0.3 pp of float64's final error) within
     F_ONE * s * sqrt(1 + 1/n_c) + (the checkpoint spacing there: the resolution of the measurement) of the controls' mean settling
     iteration, s = their sample standard deviation: a lead or lag is bounded in iterations.
  3. THE ACCEPTANCE NUMBER, on the final checkpoint and on the mean over the last third of the run:
       | mean over the HIP members - mean over the control runs |  <=  F_ENS * s_c * sqrt(1/n_c + 1/n_h) + 0.3 pp
     s_c = the controls' sample standard deviation (n_c = 4 - 5 runs incl. float64), n_h = 4, F_ENS = 2.5 (two-sided ~98 % for a
     difference of means of exchangeable runs) — the standard error of the comparison, not the range tripled; where the reference
     reproduces itself (s_c = 0: the 'k300' plateau) this IS +-0.3 pp.  And no single member further from the controls' mean than
     F_ONE * s_c * sqrt(1 + 1/n_c) + 0.3 pp (F_ONE = 3: a prediction interval for one more exchangeable run).
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
F_ENS, F_ONE = 2.5, 3.0           # factors on the controls' standard error (ensemble mean) / standard deviation (one member): module docstring
# the HIP path's own equally accurate evaluations of a trajectory: 3x3 routing x filter-gradient pixel split, on both launch paths
MEMBERS = [dict(name='default routing, library split, overlap', policy=None, split_div=1, mode='overlap'),
           dict(name='halo kernels everywhere, library split, one stream', policy=1, split_div=1, mode='eager'),
           dict(name='default routing, half split, one stream', policy=None, split_div=2, mode='eager'),
           dict(name='halo kernels everywhere, half split, overlap', policy=1, split_div=2, mode='overlap')]


def _state_of(tr):
    return {n: dict(p=s.p.cpu(), m=s.m.cpu(), v=s.v.cpu(), s=s.s.cpu(), step=s.step.cpu(), ema=None if s.ema is None else s.ema.cpu())
            for n, s in tr.cx.stores.items()}


def _load_state(tr, state):
    for n, s in tr.cx.stores.items():
        for k in ('p', 'm', 'v', 's', 'step'):
            getattr(s, k).copy_(state[n][k])
        if s.ema is not None:
            s.ema.copy_(state[n]['ema'])


def run_hip(M, fixture, policy=None, split_div=1, mode='auto'):
    """the fixture's run on the HIP path -> ({checkpoint: error rate}, per-iteration losses).  policy: tg_conv3x3_policy for the run;
    split_div: the filter gradients' pixel split (tg_wgrad_splits) divided by this — another partition of the same sums; mode:
    config.EXEC_MODE ('eager' = one stream, 'overlap' / 'auto' = filter gradients on the second stream)."""
    import torch
    from oracle import step_cifar10 as S
    from tg import geom, lib
    from tg.runtime import InjectedRNG
    lib_splits = geom.wgrad_splits
    if split_div > 1:
        geom.wgrad_splits = lambda d, bf16=False: max(1, lib_splits(d, bf16) // split_div)
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
            tr = G.fresh_trainer(G.make_config(sizes, EXEC_MODE=mode), S.init_params(0) if state is None else None)
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
        geom.wgrad_splits = lib_splits
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
def test_error_rate_ensemble_against_the_cpu_reference_runs(fixture):
    import make_golden_long as M
    ctl = M.load(fixture)
    assert len(ctl) >= 3, "a float64 run and at least two float32 controls"
    steps = [int(s) for s in ctl['f64']['eval_steps']]
    for v in ctl.values():
        assert [int(s) for s in v['eval_steps']] == steps
    cerr = {name: 1.0 - v['eval_acc'] for name, v in ctl.items()}            # variant -> errors at the checkpoints
    all_err = np.stack(list(cerr.values()))                                   # [variant, checkpoint]
    n_c = all_err.shape[0]
    runs = []
    for mb in MEMBERS:
        got, losses = run_hip(M, fixture, mb['policy'], mb['split_div'], mb['mode'])
        runs.append(dict(name=mb['name'], err=[float(got[s]) for s in steps], losses=losses))
    herr = np.asarray([r['err'] for r in runs])                               # [member, checkpoint]
    n_h = herr.shape[0]
    table = [dict(step=s, hip=[float(v) for v in herr[:, i]], **{name: float(e[i]) for name, e in cerr.items()}) for i, s in enumerate(steps)]
    tail = [i for i, s in enumerate(steps) if s > steps[-1] * 2 // 3]
    stats = {}
    for name, idx in (('final', [len(steps) - 1]), ('mean over the last third', tail)):
        c = all_err[:, idx].mean(axis=1)
        h = herr[:, idx].mean(axis=1)
        s_c = float(c.std(ddof=1))
        stats[name] = dict(controls=[float(v) for v in c], hip=[float(v) for v in h], controls_mean=float(c.mean()), hip_mean=float(h.mean()),
                           s_c=s_c, bound_ensemble=F_ENS * s_c * float(np.sqrt(1.0 / n_c + 1.0 / n_h)) + PP,
                           bound_member=F_ONE * s_c * float(np.sqrt(1.0 + 1.0 / n_c)) + PP)
    dbg = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(dbg):
        json.dump(dict(fixture=fixture, members=[r['name'] for r in runs], table=table, acceptance=stats,
                       hip_losses=[r['losses'].tolist() for r in runs]), open(os.path.join(dbg, 'long_horizon_%s.json' % fixture), 'w'), indent=1)
    level = float(cerr['f64'][-1]) + PP
    c_settle = [settling_step(steps, list(e), level) for e in cerr.values()]
    # a settling iteration is only known to the checkpoint spacing around it (k300: every 5 iterations up to 100, every 25 from there on)
    res = max(b - a for a, b in zip(steps, steps[1:]) if a <= max(c_settle) + 1 and b >= min(c_settle))
    settle_tol = F_ONE * float(np.std(c_settle, ddof=1)) * float(np.sqrt(1.0 + 1.0 / n_c)) + res
    ref = ctl['f64']['losses']
    dev = np.max([np.abs(v['losses'] - ref) for n, v in ctl.items() if n != 'f64'], axis=0)      # [iteration, 3]
    for r, he in zip(runs, herr):
        # identical weights, deterministic evaluation: the initial error is the same number (an arg-max tie at most)
        assert abs(he[0] - cerr['f64'][0]) <= 1.0 / M.N_TEST + 1e-9, (r['name'], table[0])
        # 1. per checkpoint, against the controls' range in a window of one checkpoint either side — from the iteration on by which check 2
        # requires every run to have settled (before it the trajectory is chaotic: what is asserted there is check 2's bound on the lag)
        for i, s in enumerate(steps):
            if s < np.mean(c_settle) + settle_tol:
                continue
            win = all_err[:, max(0, i - 1):i + 2]
            lo, hi = float(win.min()), float(win.max())
            w = hi - lo
            assert lo - w - PP <= he[i] <= hi + w + PP, (r['name'], 'checkpoint', s, he[i], (lo, hi), table)
        # 2. the settling iteration
        h_settle = settling_step(steps, list(he), level)
        assert abs(h_settle - np.mean(c_settle)) <= settle_tol, (r['name'], 'settling iteration', h_settle, sorted(c_settle), settle_tol, table)
        # the losses stay on the controls' scale (GAN losses fluctuate: bound = the controls' spread around float64 over the neighbouring
        # 25 iterations, per loss).  A GAN loss also SPIKES for an iteration or two and recovers — the float32 controls do (fixture 'ref':
        # D loss 1.88 against a running level of 1.05) — and the twelve sampled iterations of a run can land on one: one sampled point per
        # run may exceed the bound, none by more than three times, and the largest loss of the whole run stays within 1.5x of the controls'.
        losses = r['losses']
        assert np.isfinite(losses).all(), r['name']
        over = []
        for k in range(24, len(losses), 25):
            lo, hi = max(0, k - 25), min(len(losses), k + 25)
            allowed = 2.0 * dev[lo:hi].max(axis=0) + ref[lo:hi].std(axis=0) + 0.05
            d_k = np.abs(losses[k] - ref[k])
            assert np.all(d_k <= 3.0 * allowed), (r['name'], k + 1, losses[k], ref[k], allowed)
            if not np.all(d_k <= allowed):
                over.append((k + 1, losses[k].tolist(), ref[k].tolist(), allowed.tolist()))
        assert len(over) <= 1, (r['name'], 'sampled losses beyond the controls\' spread', over)
        c_max = np.max([np.abs(v['losses']).max(axis=0) for v in ctl.values()], axis=0)
        assert np.all(np.abs(losses).max(axis=0) <= 1.5 * c_max + 0.05), (r['name'], np.abs(losses).max(axis=0), c_max)
    # 3. the acceptance number: ensemble mean against ensemble mean, and every member against the controls' mean
    for name, st in stats.items():
        assert abs(st['hip_mean'] - st['controls_mean']) <= st['bound_ensemble'], (name, 'ensemble', st)
        for r, h in zip(runs, st['hip']):
            assert abs(h - st['controls_mean']) <= st['bound_member'], (name, r['name'], h, st)
