#!/usr/bin/env python3
"""64x64x3 / batch-256 stress configuration (BASELINE.json configs[4]): HBM-side roofline capture.

    python tools/stress64.py [--steps K] [--warmup W]        (one GPU)

Runs the D+G+C step of Model/Good_GAN_stress64.py on synthetic U(-1,1) images already resident in HBM, then one
instrumented eager iteration with per-kernel-class HIP-event timing.  Prints ONE JSON line: ms/step, images/sec and, per
kernel class, launches, milliseconds, executed GFLOP or algorithmic GB and the resulting TFLOP/s or GB/s against the
MI355X peaks.  No parity target (the configuration is not in the reference).
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

PEAK_TF, PEAK_GBS = 157.3, 8000.0


def make_config():
    from config import Config

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "stress64"
        DATA_DIR = "/nonexistent"
        NUM_LABEL = 4000
        BATCH_SIZE_G = 256
        BATCH_SIZE_L_C = 128
        BATCH_SIZE_U_C = 128
        BATCH_SIZE_L_D = 51
        BATCH_SIZE_U_D = 205
        BATCH_SIZE = 256
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 64, 64, 3
        FAKE_G_LAMBDA = 0.3
        Z_DIM = 100
        NUM_CLASSES = 10
        LEARNING_RATE = 3e-4
        CLA_LEARNINIG_RATE = 3e-3
        EPOCHS = 1
        TRAIN_SIZE = 56000
        SUMMARY = False
        USE_HIP_GRAPH = True
        SEED = 0

    return TempConfig()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    args = ap.parse_args()
    import torch
    from tg import lib
    from Training.Train_goodGAN import Train
    from Model.Good_GAN_stress64 import Good_GAN_stress64

    cfg = make_config()
    tr = Train(cfg, None, None)
    tr._build_train_graph(Good_GAN_stress64)
    tr.set_hyper(lambda_1=cfg.FAKE_G_LAMBDA, lambda_2=0.5)
    cx = tr.cx
    rng = np.random.default_rng(1234)
    img = lambda n: rng.uniform(-1, 1, (n, 64, 64, 3)).astype(np.float32)
    oh = lambda n: np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    tr.feed(dict(x_l_c=img(128), y_l_c=oh(128), x_l_d=img(51), y_l_d=oh(51), x_u_d=img(205), x_u_c=img(128)))

    def step():
        tr.sample_latent()
        tr.train_iteration()

    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    losses = tr.losses()
    assert all(np.isfinite(losses)), losses

    lib.call('tg_prof_reset')
    lib.call('tg_prof_enable', 1)
    tr.sample_latent()
    tr.train_iteration(use_graph=False)
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
            e.update(gflop=round(f.value / 1e9, 1), tflops=round(f.value / ms.value / 1e9, 1), frac_of_mfma_peak=round(f.value / ms.value / 1e9 / PEAK_TF, 3))
        if b.value > 0:
            e.update(gbytes=round(b.value / 1e9, 2), gb_per_s=round(b.value / ms.value / 1e6, 0), frac_of_hbm_peak=round(b.value / ms.value / 1e6 / PEAK_GBS, 3))
        classes[name] = e
    dump = os.environ.get('TG_PROF_DUMP')
    if dump:
        lib.call('tg_prof_dump', dump.encode())
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps({"workload": "synthetic 64x64x3, bs=256 (B_G/L_C/U_C/L_D/U_D=256/128/128/51/205), Good_GAN_stress64 D+G+C step, fp32",
                      "ms_per_step": round(dt * 1e3, 3), "images_per_sec": round(256 / dt, 1), "steps": args.steps, "hbm_gib_allocated": round(mem, 2),
                      "losses_d_g_c": [round(v, 4) for v in losses], "classes": classes}), flush=True)


if __name__ == "__main__":
    main()
