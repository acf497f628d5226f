#!/usr/bin/env python3
"""Generates tests/golden/goodgan_svhn_bf16_step_ref.npz — BASELINE.json configs[3] ("SVHN 32x32x3, 1000 labelled, bf16 MFMA conv path")
at ITS OWN batch sizes (100 / 50 / 50 / 20 / 80, Training/Train_goodGAN.py:490-496): the three solver runs of one iteration
(Training/Train_goodGAN.py:266-276; Model/Good_GAN.py:249-350 classifier, :126-206 discriminator, :35-83 generator), each started from the
SAME fixed weights, evaluated by the float64 oracle with its bf16 emulation switched on (oracle.tf_ops.MFMA_BF16: every operand of a
conv / transposed-conv / dense contraction rounded to bfloat16, products accumulated exactly).

Why a committed fixture: at these sizes the halo-tiled bf16 kernels (conv3x3_pipe_kernel<..., BF16>, wgrad3x3_kernel<..., BF16>) are the ones
the default routing takes, i.e. the launches the configs[3] step time is measured on; the oracle needs ~1 minute per solver run at these
sizes (too slow for the GPU box's test run, which only loads this file) and cannot travel as an import of the reference (TF1, SURVEY §8c:
"parity unpinned" — the vectors come from the RESTATEMENT).

Phase-ISOLATED protocol: D-update, G-update and C-update each from the initial weights P0 (no optimiser step in between) — what is compared
is every gradient the three backward passes produce, the three losses, the classifier logits that decide the discriminator's labels, and
the batch-norm moving statistics after each run.  The cross-phase flow (Adam step -> next run) is covered at small sizes by
tests/test_gpu_goodgan.py::test_synchronised_iteration.

A gradient has up to 3.3 M elements and there are 99 variables: the file keeps, per variable and solver run, its L2 norm, its largest
magnitude, SAMPLE elements at seeded positions and PROJ seeded random-sign projections (a projection of the error vector e is ~N(0, |e|^2):
16 of them bound |e| without shipping the vector).  ~0.5 MB.

    python tests/golden/make_golden_svhn_bf16_step.py          (about 3 minutes of NumPy float64 on 8 cores)
"""
import copy
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import nets_goodgan as N  # noqa: E402
from oracle import step_goodgan as S  # noqa: E402
from oracle import tf_ops as T  # noqa: E402

DATA = 'svhn'
SIZES = dict(S.SIZES['svhn'])                     # 100 / 50 / 50 / 20 / 80
HYPER = dict(lr=3e-4, cla_lr=3e-4, beta1=0.5, lambda_1=0.1, lambda_2=0.0)          # Training/Train_goodGAN.py:505; lambda_1 as late in the schedule
SEED_P, SEED_B, SEED_R = 11, 31, 32
SAMPLE, PROJ = 256, 16
PATH = os.path.join(HERE, 'goodgan_svhn_bf16_step_ref.npz')


def f64(d):
    return {k: (f64(v) if isinstance(v, dict) else np.asarray(v, np.float64)) for k, v in d.items()}


def init_params():
    """float32 initial weights (tests/test_oracle_goodgan.py::scrambled: the reference's shapes, benign values, non-trivial statistics)."""
    from test_oracle_goodgan import scrambled
    return {k: v.astype(np.float32) for k, v in scrambled(DATA, SEED_P).items()}


def inputs():
    return S.synth_batch(DATA, SEED_B, SIZES), S.synth_rnd(DATA, SEED_R, SIZES)


def _rng_of(name):
    return np.random.default_rng(zlib.crc32(name.encode()))


def summary(name, g):
    """the part of gradient `g` of variable `name` the fixture keeps (module docstring); the SAME function digests the HIP path's gradient."""
    v = np.asarray(g, np.float64).reshape(-1)
    rng = _rng_of(name)
    idx = rng.choice(v.size, min(SAMPLE, v.size), replace=False)
    proj = np.empty(PROJ)
    for j in range(PROJ):
        proj[j] = float(np.dot(rng.integers(0, 2, v.size).astype(np.float64) * 2.0 - 1.0, v))
    return dict(l2=float(np.linalg.norm(v)), amax=float(np.abs(v).max()), sample=v[idx], proj=proj)


def run():
    T.MFMA_BF16 = True
    try:
        P0 = init_params()
        b, rnd = inputs()
        b64, r64 = f64(b), f64(rnd)
        st0 = S.new_state(f64(P0))
        out = {}
        for phase, fn, net in (('D', S.d_phase, 'discriminator'), ('G', S.g_phase, 'good_generator'), ('C', S.c_phase, 'classifier')):
            t0 = time.time()
            st = copy.deepcopy(st0)
            out['loss/' + phase] = np.float64(fn(st, DATA, b64, r64[phase], HYPER))
            for k, g in st['last_grads'][phase].items():
                for what, val in summary(k, g).items():
                    out['grad/%s/%s/%s' % (phase, k, what)] = np.asarray(val)
            for k, v in st['P'].items():                 # batch-norm moving statistics this solver run advanced (small vectors, kept whole)
                if 'moving_' in k and not np.array_equal(v, st0['P'][k]):
                    out['stat/%s/%s' % (phase, k)] = np.asarray(v)
            if phase == 'D':
                out['d_labels_logits/unl'], out['d_labels_logits/unl_d'] = st['last_logits']['unl'], st['last_logits']['unl_d']
            if phase == 'C':                             # the labels of D(x_u_c): arg-max of C_unl's logits of THIS run
                bnu = {}
                c_unl = N.seq_fwd(st0['P'], N.classifier_layers(DATA), b64['x_u_c'], None, r64['C']['C_unl'], True, bnu)[0]
                out['c_labels_logits/unl'] = c_unl
            print('%s-update: %.0f s, loss %.6f' % (phase, time.time() - t0, float(out['loss/' + phase])), flush=True)
        return out
    finally:
        T.MFMA_BF16 = False


def load():
    return dict(np.load(PATH))


if __name__ == "__main__":
    g = run()
    np.savez_compressed(PATH, **{k: (np.asarray(v, np.float32) if k.endswith(('/sample',)) or k.startswith(('stat/', 'd_labels', 'c_labels')) else np.asarray(v))
                                 for k, v in g.items()})
    print('wrote', PATH, os.path.getsize(PATH), 'bytes')
