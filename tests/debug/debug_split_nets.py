"""Diagnostic: classifier forward / backward on five images in two applications (tests/test_gpu_nets.py::test_classifier_fwd_bwd_two_segments)
— dumps every variable gradient; run once with TG_IGEMM_NOSPLIT=1 and once without, then compare (second argument: the other dump)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tensorflow-implementation-of-triple-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import gpu_common as G
from oracle import step_cifar10 as S
from tg.runtime import InjectedRNG
out = sys.argv[1]
P = S.init_params(0)
rng0 = np.random.default_rng(100)                       # tests/test_gpu_nets.py::scrambled_params(0)
for k in P:
    if k.endswith(('/g', 'gamma')):
        P[k] = (1 + 0.3 * rng0.standard_normal(P[k].shape)).astype(np.float32)
    elif k.endswith(('/b', 'bias', 'beta')):
        P[k] = (0.1 * rng0.standard_normal(P[k].shape)).astype(np.float32)
tr = G.fresh_trainer(G.make_config(dict(B_G=6, L_C=3, U_C=2, L_D=2, U_D=4)), P)
cx, m = tr.cx, tr.model
sizes = dict(S.SIZES, L_C=3, U_C=2)
rnd = S.synth_rnd(1, sizes)
b = S.synth_batch(2, sizes)
cx.rng = InjectedRNG({'T/C/' + k: v for k, v in G.cat_rnd(rnd['C']['C_real'], rnd['C']['C_unl']).items()}, cx.device)
dl = np.random.default_rng(3).standard_normal((5, 10)).astype(np.float32)
with cx.phase_scope('T', train_nets=('classifier',)):
    xa = cx.from_numpy(np.concatenate([b['x_l_c'], b['x_u_c']]))
    with cx.rng_scoped('T/C'):
        logits, feat = m.classifier(xa, True, segments=[3, 2])
    logits.grad = cx.from_numpy(dl, ld=32)
    cx.backward()
st = cx.stores['classifier']
d = {k.replace('/', '.'): st.get(k, 'grad') for k in st.names(True)}
np.savez(out, **d)
if len(sys.argv) > 2:
    o = np.load(sys.argv[2])
    for k in d:
        df = np.abs(d[k] - o[k])
        e = df.max() / (np.abs(o[k]).max() + 1e-30)
        print('%-40s rel diff %.3e  elements off by > 1e-4 of max: %d of %d %s' % (k, e, int((df > 1e-4 * np.abs(o[k]).max()).sum()), df.size, '<--' if e > 1e-3 else ''))
