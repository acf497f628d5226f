import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from oracle import step_cifar10 as S, nets_cifar10 as N, tf_ops as T
import gpu_common as G
import test_gpu_step as TS
from tg.runtime import InjectedRNG

sizes = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)
hyper = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)
st, tr, zca = TS.setup(sizes, hyper)
full = dict(S.SIZES, **sizes)
batch, rnd = S.synth_batch(100, full), S.synth_rnd(200, full)
b64, r64 = TS.f64(batch), TS.f64(rnd)
tr.cx.rng = InjectedRNG(G.injected_arrays(rnd), tr.cx.device)
tr.feed(batch)
c_ref = S.c_phase(st, b64, r64['C'], hyper, zca)
tr._c_forward_backward()
store = tr.cx.stores['classifier']
for k, gref in st['last_grads']['C'].items():
    got = store.get(k, 'grad')
    print('%-40s err %.3e  max %.3e  rel %.2e' % (k, np.abs(got - gref).max(), np.abs(gref).max(), np.abs(got - gref).max() / (np.abs(gref).max() + 1e-30)))
print('losses', c_ref, tr.losses()[2])
