"""Diagnostic (GPU + oracle): where does the HIP path's free-running long-horizon trajectory part from the float64 oracle's?
(round-2 verdict item 1a: error rate 68.7 % against the golden 5.7 % at iteration 50 of tests/golden/make_golden_long.py 'k300').

The first N iterations of that exact run, PHASE-SYNCHRONISED (tests/test_gpu_step.py::run_synchronised: every solver run starts from
the oracle's float64 weights copied into the HIP stores), each solver run evaluated TWICE on the HIP path — filter gradients of the
classifier's 3x3 layers on csrc/wgrad3x3.hip (tg_conv3x3_policy 0, default) and on the generic per-tap kernel (policy 2) — and for
every variable the relative L2 error of the pre-Adam gradient against the float64 oracle is recorded.  If one routing carries a
systematic term it shows as a per-variable error that is larger for that routing at every iteration; if both sit at the same
rounding-level error the free-running gap is trajectory divergence (chaos), which the float32 controls of the fixture quantify.

    python tests/debug/debug_long_horizon_lag.py [N=50] > gpurun_out/lag.log      -> gpurun_out/long_horizon_lag.json
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tensorflow-implementation-of-triple-gan_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np
import torch

import gpu_common as G
import make_golden_long as M
import test_gpu_step as TS
from oracle import step_cifar10 as S
from tg import lib
from tg.runtime import InjectedRNG

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NETS = TS.NETS
hyper = M.HYPER
st, tr, zca = TS.setup(M.SIZES, hyper)
cx, stores = tr.cx, tr.cx.stores
rows = []


def snapshot():
    return {n: s.s.clone() for n, s in stores.items()}, getattr(tr, '_g_saved', None)


def restore(snap):
    for n, t in snap[0].items():
        stores[n].s.copy_(t)
    tr._g_saved = snap[1]


def grad_errors(key):
    out = {}
    store = stores[NETS[key]]
    for k, gref in st['last_grads'][key].items():
        d = store.get(k, 'grad').astype(np.float64) - gref
        out[k] = float(np.linalg.norm(d) / (np.linalg.norm(gref) + 1e-30))
    return out


def both_routings(key, fn):
    """run solver run `key` under both policies from the same state; leaves the default routing's result in the stores."""
    snap = snapshot()
    res = {}
    for policy in (2, 0):
        restore(snap)
        was = lib.call('tg_conv3x3_policy', policy)
        fn()
        torch.cuda.synchronize()
        lib.call('tg_conv3x3_policy', was)
        res['generic' if policy == 2 else 'wgrad3x3'] = grad_errors(key)
    return res


t0 = time.time()
for it in range(N):
    batch, rnd = M.inputs(it)
    b64, r64 = TS.f64(batch), TS.f64(rnd)
    cx.rng = InjectedRNG(G.injected_arrays(rnd), cx.device)
    tr.feed(batch)
    row = dict(it=it)
    S.d_phase(st, b64, r64['D'], hyper, zca)
    row['D'] = both_routings('D', tr._d_forward_backward)
    tr._train_op(tr.d_optimizer, stores['discriminator'])
    for k in stores['discriminator'].names():
        if 'moving_' not in k:
            stores['discriminator'].set(k, st['P'][k])
    TS.sync_pop_means(st, tr, check=False)
    S.g_phase(st, b64, r64['G'], hyper)
    row['G'] = both_routings('G', tr._g_forward_backward)
    tr._train_op(tr.g_optimizer, stores['good_generator'])
    for k in stores['good_generator'].names():
        if 'moving_' not in k:
            stores['good_generator'].set(k, st['P'][k])
    S.c_phase(st, b64, r64['C'], hyper, zca)
    row['C'] = both_routings('C', tr._c_forward_backward)
    tr._c_apply()
    for k in stores['classifier'].names():
        stores['classifier'].set(k, st['P'][k])
    for key, net in NETS.items():                       # optimiser slots follow the oracle too
        store = stores[net]
        for k in store.names(True):
            kind, off, n, shape = store.index[k]
            store.m[off:off + n].copy_(torch.from_numpy(st['m'][k].astype(np.float32).reshape(-1)))
            store.v[off:off + n].copy_(torch.from_numpy(st['v'][k].astype(np.float32).reshape(-1)))
    rows.append(row)
    worst = {r: max(row['C'][r].items(), key=lambda kv: kv[1]) for r in row['C']}
    print('it %d (%.0f s)  worst classifier gradient error  %s' % (it, time.time() - t0, {r: (k, '%.2e' % v) for r, (k, v) in worst.items()}), flush=True)

out = os.path.join(ROOT, 'gpurun_out', 'long_horizon_lag.json')
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out, 'w'))
# summary: per variable, geometric-mean error over the iterations, both routings side by side
print('\nvariable, geometric mean of the relative L2 gradient error over %d iterations: wgrad3x3 | generic' % N)
for key in 'DGC':
    for k in rows[0][key]['generic']:
        a = np.exp(np.mean([np.log(r[key]['wgrad3x3'][k] + 1e-30) for r in rows]))
        b = np.exp(np.mean([np.log(r[key]['generic'][k] + 1e-30) for r in rows]))
        print('%s %-60s %.3e | %.3e %s' % (key, k, a, b, '  <-- differs' if max(a, b) > 2 * min(a, b) else ''))
