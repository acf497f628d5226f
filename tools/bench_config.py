#!/usr/bin/env python3
"""Step time of the BASELINE.json configurations that are NOT the bench.py line (one GPU):

    python tools/bench_config.py --config stress64   synthetic 64x64x3, bs 256, deeper G/D/C (configs[4]; HBM-side roofline capture)
    python tools/bench_config.py --config svhn-bf16  SVHN 32x32x3, Good_GAN svhn, bf16 MFMA conv path (configs[3])
    python tools/bench_config.py --config svhn       the same in fp32
    python tools/bench_config.py --config mnist      MNIST 28x28x1, Good_GAN mnist (configs[0] shape)
    python tools/bench_config.py --config cifar10[-bf16]   the bench.py workload (Good_GAN_cifar10, synthetic ZCA), fp32 / bf16 operands

Runs the D+G+C step on synthetic images already resident in HBM, then one instrumented eager iteration with
per-kernel-class HIP-event timing.  Prints ONE JSON line: ms/step, images/sec and, per kernel class, launches,
milliseconds, executed GFLOP or algorithmic GB and the resulting TFLOP/s or GB/s against the MI355X peaks of the arithmetic the
launches use (fp32 MFMA 157.3 TFLOP/s; bf16 MFMA 2 500 TFLOP/s dense for the *-bf16 configurations — MI355X_MICROARCH.md) and 8 TB/s HBM.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

PEAK_TF, PEAK_TF_BF16, PEAK_GBS = 157.3, 2500.0, 8000.0


SHAPES = {   # name: (data, H, C, B_G, L_C, U_C, L_D, U_D, mfma dtype, lambda_1, lr, cla_lr)
    'stress64': ('stress64', 64, 3, 256, 128, 128, 51, 205, 'f32', 0.3, 3e-4, 3e-3),
    'svhn': ('svhn', 32, 3, 100, 50, 50, 20, 80, 'f32', 0.1, 3e-4, 3e-4),
    'svhn-bf16': ('svhn', 32, 3, 100, 50, 50, 20, 80, 'bf16', 0.1, 3e-4, 3e-4),
    'mnist': ('mnist', 28, 1, 100, 100, 100, 20, 80, 'f32', 0.1, 1e-3, 3e-4),
    'cifar10': ('cifar10', 32, 3, 100, 50, 50, 20, 80, 'f32', 0.3, 3e-4, 3e-3),
    'cifar10-bf16': ('cifar10', 32, 3, 100, 50, 50, 20, 80, 'bf16', 0.3, 3e-4, 3e-3),
}


def make_config(name='stress64'):
    from config import Config
    data, hw, ch, bg, lc, uc, ld, ud, prec, lam, lr, clr = SHAPES[name]

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = data
        DATA_DIR = "/nonexistent"
        NUM_LABEL = 4000
        BATCH_SIZE_G = bg
        BATCH_SIZE_L_C = lc
        BATCH_SIZE_U_C = uc
        BATCH_SIZE_L_D = ld
        BATCH_SIZE_U_D = ud
        BATCH_SIZE = bg
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = hw, hw, ch
        FAKE_G_LAMBDA = lam
        Z_DIM = 100
        NUM_CLASSES = 10
        MINIBATCH_DIS = False
        LEARNING_RATE = lr
        CLA_LEARNINIG_RATE = clr
        EPOCHS = 1
        TRAIN_SIZE = 56000
        SUMMARY = False
        USE_HIP_GRAPH = None
        EXEC_MODE = os.environ.get('TG_EXEC_MODE', 'auto')
        SEED = 0
        MFMA_DTYPE = prec

    return TempConfig()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=sorted(SHAPES), default='stress64')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=5)
    args = ap.parse_args()
    import torch
    from tg import lib
    from Training.Train_goodGAN import Train
    if args.config == 'stress64':
        from Model.Good_GAN_stress64 import Good_GAN_stress64 as Model
    elif args.config.startswith('cifar10'):
        from Model.Good_GAN_cifar10 import Good_GAN_cifar10 as Model
    else:
        from Model.Good_GAN import Good_GAN as Model

    cfg = make_config(args.config)
    if args.config.startswith('cifar10'):
        q, _ = np.linalg.qr(np.random.default_rng(4321).standard_normal((3072, 3072)))      # SURVEY §8d synthetic whitening
        cfg.ZCA = (np.zeros(3072, np.float32), q.astype(np.float32))
    tr = Train(cfg, None, None)
    tr._build_train_graph(Model)
    tr.set_hyper(lambda_1=cfg.FAKE_G_LAMBDA, lambda_2=0.5)
    cx = tr.cx
    rng = np.random.default_rng(1234)
    lo = 0.0 if cfg.DATA_NAME == 'mnist' else -1.0
    img = lambda n: rng.uniform(lo, 1, (n, cfg.IMAGE_HEIGHT, cfg.IMAGE_WIDTH, cfg.CHANNEL)).astype(np.float32)
    oh = lambda n: np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    tr.feed(dict(x_l_c=img(cfg.BATCH_SIZE_L_C), y_l_c=oh(cfg.BATCH_SIZE_L_C), x_l_d=img(cfg.BATCH_SIZE_L_D), y_l_d=oh(cfg.BATCH_SIZE_L_D),
                 x_u_d=img(cfg.BATCH_SIZE_U_D), x_u_c=img(cfg.BATCH_SIZE_U_C)))

    def step():
        tr.sample_latent()
        tr.train_iteration()

    for _ in range(max(args.warmup, 2) if cfg.EXEC_MODE != 'auto' else max(args.warmup, tr.AUTO_ITERS + 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = (time.perf_counter() - t0) / args.steps     # host time to ISSUE an iteration (the queue is drained only below): when this is
    torch.cuda.synchronize()                              # close to the step time, the step is bound by the host's launch rate
    dt = (time.perf_counter() - t0) / args.steps
    issue = []                                            # the launch path's own cost: ONE iteration issued into an empty queue (t_issue above includes
    for _ in range(5):                                    # the time the host is blocked on a full queue once it runs ahead of the GPU)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        issue.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    t_issue_free = float(np.median(issue))
    losses = tr.losses()
    assert all(np.isfinite(losses)), losses

    lib.call('tg_prof_reset')
    lib.call('tg_prof_enable', 1)
    tr.sample_latent()
    mode_was, cfg.EXEC_MODE = cfg.EXEC_MODE, 'eager'            # one stream: HIP events bracket one kernel
    tr.train_iteration(use_graph=False)
    cfg.EXEC_MODE = mode_was
    torch.cuda.synchronize()
    lib.call('tg_prof_enable', 0)
    classes = {}
    for cls in range(lib.call('tg_prof_num_classes')):
        ms, n, f, b = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        lib.call('tg_prof_collect', cls, C.byref(ms), C.byref(n), C.byref(f), C.byref(b))
        name = lib.load().tg_prof_class_name(cls).decode()
        if n.value == 0:
            continue
        e = dict(launches=n.value, ms=round(ms.value, 3))
        if f.value > 0:
            peak = PEAK_TF_BF16 if cfg.MFMA_DTYPE == 'bf16' else PEAK_TF             # the MFMA classes run in the configuration's operand type
            e.update(gflop=round(f.value / 1e9, 1), tflops=round(f.value / ms.value / 1e9, 1), frac_of_mfma_peak=round(f.value / ms.value / 1e9 / peak, 4),
                     mfma_peak_tflops=peak)
        if b.value > 0:
            e.update(gbytes=round(b.value / 1e9, 2), gb_per_s=round(b.value / ms.value / 1e6, 0), frac_of_hbm_peak=round(b.value / ms.value / 1e6 / PEAK_GBS, 3))
        classes[name] = e
    dump = os.environ.get('TG_PROF_DUMP') or os.path.join('/tmp', 'tg_prof_cfg_%d.csv' % os.getpid())
    lib.call('tg_prof_dump', dump.encode())
    # the three longest MFMA launches of the iteration against BOTH roofs: executed FLOP / the operand type's dense MFMA peak, and algorithmic
    # bytes (operands once + output once, fp32 tensors in HBM) / 8 TB/s — with bf16 operands the 3x3 layers are bound by the second
    import csv
    peak = PEAK_TF_BF16 if cfg.MFMA_DTYPE == 'bf16' else PEAK_TF
    rows = [r for r in csv.DictReader(open(dump)) if r['class'] in ('igemm_f32', 'wgrad_f32') and float(r['ms']) > 0]
    rows.sort(key=lambda r: -float(r['ms']))
    largest = [dict(kind=r['class'], shape=r['desc'], ms=round(float(r['ms']), 4), tflops=round(float(r['gflop']) / float(r['ms']), 1),
                    frac_of_mfma_peak=round(float(r['gflop']) / float(r['ms']) / peak, 4), gb_per_s=round(float(r['gbytes']) / float(r['ms']) * 1e3, 0),
                    frac_of_hbm_peak=round(float(r['gbytes']) / float(r['ms']) * 1e3 / PEAK_GBS, 3)) for r in rows[:3]]
    if not os.environ.get('TG_PROF_DUMP'):
        os.remove(dump)
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps({"workload": "%s: synthetic %dx%dx%d, B_G/L_C/U_C/L_D/U_D=%d/%d/%d/%d/%d, %s D+G+C step, MFMA operands %s" % (
                          args.config, cfg.IMAGE_HEIGHT, cfg.IMAGE_WIDTH, cfg.CHANNEL, cfg.BATCH_SIZE_G, cfg.BATCH_SIZE_L_C, cfg.BATCH_SIZE_U_C,
                          cfg.BATCH_SIZE_L_D, cfg.BATCH_SIZE_U_D, Model.__name__, cfg.MFMA_DTYPE),
                      "ms_per_step": round(dt * 1e3, 3), "images_per_sec": round(cfg.BATCH_SIZE_G / dt, 1), "steps": args.steps, "hbm_gib_allocated": round(mem, 2),
                      "host_issue_ms_per_step": round(t_issue_free * 1e3, 3), "host_issue_ms_per_step_queue_full": round(t_issue * 1e3, 3),
                      "largest_mfma_launches": largest, "exec_mode": cfg.EXEC_MODE, "exec_mode_chosen": tr.exec_mode_chosen()[0] if cfg.EXEC_MODE == 'auto' else cfg.EXEC_MODE,
                      "exec_mode_timings_ms": {k: round(v * 1e3, 3) for k, v in tr.exec_mode_chosen()[1].items()},
                      "losses_d_g_c": [round(v, 4) for v in losses], "classes": classes}), flush=True)


if __name__ == "__main__":
    main()
