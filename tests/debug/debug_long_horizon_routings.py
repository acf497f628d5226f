"""Diagnostic: the long-horizon fixtures on the HIP path with the 3x3 layers on the halo-tiled kernels (default policy) and on the generic
implicit-GEMM kernel (tg_conv3x3_policy(2): never) — two float32 summation orders of the same path — beside the oracle's variants.
    python tests/debug/debug_long_horizon_routings.py [fixture ...]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tensorflow-implementation-of-triple-gan_amd')):
    sys.path.insert(0, p)
import make_golden_long as M
import test_gpu_long_horizon as T

for fixture in (sys.argv[1:] or ['ref']):
    ctl = M.load(fixture)
    steps = [int(s) for s in ctl['f64']['eval_steps']]
    runs = {}
    for name, policy in (('halo kernels', None), ('generic kernel', 2)):
        err, _ = T.run_hip(M, fixture, policy)
        runs[name] = [err[s] for s in steps]
    print('fixture', fixture)
    print('%-6s %s | %s' % ('step', ' '.join('%-15s' % n for n in runs), ' '.join('%-6s' % n for n in ctl)))
    for i, s in enumerate(steps):
        print('%-6d %s | %s' % (s, ' '.join('%-15.3f' % runs[n][i] for n in runs), ' '.join('%-6.3f' % (1.0 - ctl[n]['eval_acc'][i]) for n in ctl)))
