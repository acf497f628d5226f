"""The reference's end-of-epoch "training statistics" run (Training/Train_goodGAN.py:280-293): after the last iteration of an epoch ONE
forward-only session call evaluates d_loss, g_loss and c_loss together on the last feed with train_ph = True — fresh dropout / noise
draws (a single set, shared by the three losses), one more pop_mean update of EVERY classifier application of Model.forward_pass
(C_real, C_unl, C_unl_rep, C_unl_d, C_fake) and one more moving-statistics update of the generator's batch norms — and THOSE losses are
what is logged.  Train.training_statistics() against oracle/forward_pass.py::training_statistics_cifar10 after a short synchronised epoch,
and Train.train() logging exactly that pass."""
import numpy as np
import pytest

import gpu_common as G
import test_gpu_step as TS
from oracle import forward_pass as OF
from oracle import step_cifar10 as S

pytestmark = pytest.mark.gpu
SIZES = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)
HYPER = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)


def test_epoch_tail_statistics_pass_matches_the_oracle():
    from tg.runtime import InjectedRNG
    st, tr = TS.run_synchronised(SIZES, 2, HYPER)             # a short epoch: two iterations, HIP state == oracle state afterwards
    cx, stores = tr.cx, tr.cx.stores
    for k in stores['good_generator'].names(False):          # (run_synchronised leaves the moving statistics checked, not copied)
        stores['good_generator'].set(k, st['P'][k])
    full = dict(S.SIZES, **SIZES)
    b = S.synth_batch(100 + 1, full)                          # the epoch's last feed is still in the placeholders
    r = S.synth_rnd(777, full)
    rnd = dict(C_real=r['C']['C_real'], C_unl=r['C']['C_unl'], C_unl_rep=r['C']['C_unl_rep'], C_unl_d=r['D']['C_unl_d'], C_fake=r['C']['C_fake'],
               D_real=r['D']['D_real'], D_fake=r['D']['D_fake'], D_unl=r['D']['D_unl'])
    zca = tuple(np.asarray(a, np.float64) for a in G.zca())
    before = {k: v.copy() for k, v in st['P'].items()}
    ref = OF.training_statistics_cifar10(st['P'], TS.f64(b), TS.f64(rnd), zca, [HYPER['lambda_1'], HYPER['lambda_2']])
    inj = {}
    for k, v in G.cat_rnd(rnd['C_real'], rnd['C_unl'], rnd['C_unl_rep'], rnd['C_unl_d'], rnd['C_fake']).items():
        inj['stats/C/' + k] = v
    for k, v in G.cat_rnd(rnd['D_real'], rnd['D_fake'], rnd['D_unl']).items():
        inj['stats/D/' + k] = v
    cx.rng = InjectedRNG(inj, cx.device)
    p_before = {n: s.p.clone() for n, s in stores.items()}
    steps_before = {n: int(s.step.item()) for n, s in stores.items()}
    got = tr.training_statistics()
    for a, e in zip(got, ref):
        assert abs(a - e) <= 2e-4 * max(1.0, abs(e)), (got, ref)
    last = tr.losses()                                        # the last iteration's losses are different numbers
    assert max(abs(a - l) for a, l in zip(got, last)) > 1e-4, (got, last)
    moved = 0
    for net, s in stores.items():
        assert bool((s.p == p_before[net]).all()) and int(s.step.item()) == steps_before[net]      # forward-only: nothing is trained
        for k in s.names(False):                              # ... but every running statistic advanced by this one run
            refv = st['P'][k]
            assert np.abs(s.get(k) - refv).max() <= 2e-4 * max(1.0, np.abs(refv).max()), k
            moved += int(np.abs(refv - before[k]).max() > 0)
    assert moved == 10 + 6                                    # ten pop_mean vectors, three generator batch norms x (mean, variance)


def test_train_logs_the_statistics_pass(tmp_path, monkeypatch):
    from tg import runtime
    from Training import Train_goodGAN as TG
    runtime.set_context(None)
    seen = []
    orig = TG.Train.training_statistics

    def spy(self):
        out = orig(self)
        seen.append((out, self.losses()))
        return out
    monkeypatch.setattr(TG.Train, 'training_statistics', spy)
    monkeypatch.setattr(TG, "_root_dir", lambda: str(tmp_path))

    class Flags(object):
        train_size = 4000 + 300
        sample_dir = None
        seed = 1
        summary = False
    hist = TG._main_training_cifar10(Flags(), epochs=1)
    assert len(seen) == 1
    stats, last_iteration = seen[0]
    assert (hist[0]['d_loss'], hist[0]['g_loss'], hist[0]['c_loss']) == stats and stats != last_iteration
