"""oracle/forward_pass.py (the reference's whole-graph route: Model.forward_pass -> Train_base._loss_GAN) against oracle/step_*.py
(the per-solver restatement the GPU step tests use): two independently written paths through the same networks must give the
same three losses when the applications a solver run executes are fed the same draws.  CPU, float64, tiny batches."""
import copy

import numpy as np
import pytest

from oracle import forward_pass as OF
from oracle import nets_goodgan as NG
from oracle import step_cifar10 as S
from oracle import step_goodgan as SG

SIZES = dict(B_G=3, L_C=2, U_C=2, L_D=1, U_D=2)
HYPER = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)


def f64(d):
    return {k: (f64(v) if isinstance(v, dict) else np.asarray(v, np.float64)) for k, v in d.items()}


def test_cifar10_whole_graph_losses_equal_the_solver_runs():
    full = dict(S.SIZES, **SIZES)
    P, b, r = f64(S.init_params(1)), f64(S.synth_batch(3, full)), f64(S.synth_rnd(4, full))
    zca = tuple(np.asarray(a, np.float64) for a in S.synth_zca())
    Y = [b['y_g'], b['y_l_c']]
    lam = [HYPER['lambda_1'], HYPER['lambda_2']]
    any_c = r['C']['C_real']
    # D-update's view: C_unl / C_unl_d and the three D applications draw what d_phase draws
    rd = dict(C_real=any_c, C_unl=r['D']['C_unl'], C_unl_rep=r['C']['C_unl_rep'], C_unl_d=r['D']['C_unl_d'], C_fake=r['C']['C_fake'],
              D_real=r['D']['D_real'], D_fake=r['D']['D_fake'], D_unl=r['D']['D_unl'])
    (_, D, C), _ = OF.forward_pass_cifar10(P, b, rd, zca, True)
    d_loss = OF.loss_gan(D, C, Y, lam, True)[0]
    assert abs(d_loss - S.d_phase(S.new_state(copy.deepcopy(P)), b, r['D'], HYPER, zca)) < 1e-12
    # G-update's view
    rg = dict(rd, D_fake=r['G']['D_fake'])
    (_, D, C), _ = OF.forward_pass_cifar10(P, b, rg, zca, True)
    assert abs(OF.loss_gan(D, C, Y, lam, True)[1] - S.g_phase(S.new_state(copy.deepcopy(P)), b, r['G'], HYPER)) < 1e-12
    # C-update's view
    rc = dict(C_real=r['C']['C_real'], C_unl=r['C']['C_unl'], C_unl_rep=r['C']['C_unl_rep'], C_unl_d=r['D']['C_unl_d'], C_fake=r['C']['C_fake'],
              D_real=r['D']['D_real'], D_fake=r['D']['D_fake'], D_unl=r['C']['D_unl'])
    (G_, D, C), pops = OF.forward_pass_cifar10(P, b, rc, zca, True)
    assert abs(OF.loss_gan(D, C, Y, lam, True)[2] - S.c_phase(S.new_state(copy.deepcopy(P)), b, r['C'], HYPER, zca)) < 1e-12
    assert G_.shape == (3, 32, 32, 3) and [d.shape for d in D] == [(3, 1)] * 4 + [(2, 1)] * 2 and len(C) == 5
    assert all(np.allclose(D[i], 1 / (1 + np.exp(-D[i + 1]))) for i in (0, 2, 4))
    assert len(pops) == 10                                            # every mean-only-BN layer updated its pop_mean (five times each)


@pytest.mark.parametrize("data", ['mnist', 'svhn'])
def test_goodgan_whole_graph_losses_equal_the_solver_runs(data):
    P, b, r = f64(NG.init_params(data, 2)), f64(SG.synth_batch(data, 5, SIZES)), f64(SG.synth_rnd(data, 6, SIZES))
    hyper = dict(HYPER, lambda_1=0.1)
    Y = [b['y_g'], b['y_l_c']]
    rd = dict(C_real=r['C']['C_real'], C_unl=r['D']['C_unl'], C_unl_d=r['D']['C_unl_d'], C_fake=r['C']['C_fake'],
              D_real=r['D']['D_real'], D_fake=r['D']['D_fake'], D_unl=r['D']['D_unl'])
    (_, D, C), _ = OF.forward_pass_goodgan(P, data, b, rd, True)
    assert abs(OF.loss_gan(D, C, Y, [0.1], False)[0] - SG.d_phase(SG.new_state(copy.deepcopy(P)), data, b, r['D'], hyper)) < 1e-12
    (_, D, C), _ = OF.forward_pass_goodgan(P, data, b, dict(rd, D_fake=r['G']['D_fake']), True)
    assert abs(OF.loss_gan(D, C, Y, [0.1], False)[1] - SG.g_phase(SG.new_state(copy.deepcopy(P)), data, b, r['G'], hyper)) < 1e-12
    rc = dict(rd, C_unl=r['C']['C_unl'], D_unl=r['C']['D_unl'])
    (_, D, C), _ = OF.forward_pass_goodgan(P, data, b, rc, True)
    assert abs(OF.loss_gan(D, C, Y, [0.1], False)[2] - SG.c_phase(SG.new_state(copy.deepcopy(P)), data, b, r['C'], hyper)) < 1e-12
    assert len(C) == 4
