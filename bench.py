#!/usr/bin/env python3
"""Triple-GAN training throughput on MI355X: images/sec of the full D+G+C step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts the N ranks itself, tg/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one iteration of Training/Train_goodGAN.py:266-276 (D-update, G-update, C-update + EMA) on the CIFAR-10
config of the reference (32x32x3, B_G/L_C/U_C/L_D/U_D = 100/50/50/20/80, fp32), synthetic class-prototype batches
already resident in HBM, random-init weights, synthetic orthogonal ZCA.  images/sec = BATCH_SIZE(100) * steps/sec *
replicas (weak scaling: every replica runs the single-GPU batch; gradients are sum-all-reduced over RCCL).

The JSON line also carries
  roofline     — the classifier's 3x3 convolution path (87 % of the step's FLOPs; the north-star kernel path): algorithmic
                 FLOPs of its forward / input-gradient (conv3x3_pipe_kernel, igemm_f32_kernel) and filter-gradient (wgrad3x3_kernel,
                 wgrad_f32_kernel) launches in one iteration / their summed
                 HIP-event durations, measured in an instrumented eager ONE-STREAM pass right after the timed
                 region (the timed region replays a two-stream launch plan — or hipGraphs — where a kernel's events would also
                 bracket its neighbours);
                 peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).  The figure over ALL igemm / wgrad launches
                 (generator, discriminator, dense, ZCA included) is reported next to it.
  cpu_baseline — the NumPy oracle (oracle/step_cifar10.py, a port: TF1 is not installable) timed on this host's cores:
                 1 warm-up + 3 timed iterations of the same workload, median (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3
TRAFFIC_SOURCES = ('igemm.hip', 'conv3x3_bf16.hip', 'wgrad3x3.hip')      # the kernels the roofline object names (tools/make_traffic_json.py)
SIZES = dict(B_G=100, L_C=50, U_C=50, L_D=20, U_D=80)


def algorithmic_flops(s=SIZES):
    """SURVEY §8d counting rule: conv/deconv/dense forward = 2*MACs, backward-data = backward-weight = forward, no
    data-gradient for a network's first layer unless its input needs one; only what each solver run executes."""
    mf = lambda h, k, ci, co: 2.0 * h * h * k * k * ci * co
    c_layers = [mf(32, 3, 3, 128), mf(32, 3, 128, 128), mf(32, 3, 128, 128), mf(16, 3, 128, 256), mf(16, 3, 256, 256),
                mf(16, 3, 256, 256), mf(6, 3, 256, 512), mf(6, 1, 512, 256), mf(6, 1, 256, 128), 2.0 * 128 * 10]
    d_layers = [mf(32, 3, 13, 32), mf(16, 3, 42, 32), mf(16, 3, 42, 64), mf(8, 3, 74, 64), mf(8, 3, 74, 128), mf(8, 3, 138, 128),
                2.0 * 138]
    g_layers = [2.0 * 110 * 8192, 2.0 * 4 * 4 * 25 * 522 * 256, 2.0 * 8 * 8 * 25 * 266 * 128, 2.0 * 16 * 16 * 25 * 138 * 3]
    zca = 2.0 * 3072 * 3072
    C, D, Gn = sum(c_layers), sum(d_layers), sum(g_layers)
    n_d = s['L_D'] + s['U_D'] + s['B_G'] + s['U_C']
    n_cd = s['U_C'] + s['U_D']
    n_c = s['L_C'] + 2 * s['U_C'] + s['B_G']
    fwd = {
        'D': s['B_G'] * Gn + n_cd * (zca + C) + n_d * D,
        'G': s['B_G'] * (Gn + D),          # reference work; the build reuses the D-update's generator forward (see executed_*)
        'C': s['B_G'] * Gn + (s['L_C'] + s['U_C'] + s['B_G']) * zca + n_c * C + s['U_C'] * D,
    }
    dgrad = {'D': n_d * (D - d_layers[0]), 'G': s['B_G'] * (D + Gn - g_layers[0]), 'C': n_c * (C - c_layers[0])}
    wgrad = {'D': n_d * D, 'G': s['B_G'] * Gn, 'C': n_c * C}
    igemm = sum(fwd.values()) + sum(dgrad.values())
    wg = sum(wgrad.values())
    conv3x3 = n_c * sum(c_layers[:7])     # classifier 3x3 convolutions, forward
    reused = s['B_G'] * Gn                 # generator forward of the G-update: same feed, same weights as in the D-update -> kept
    return dict(total=igemm + wg, igemm=igemm, wgrad=wg, c_conv3x3_fwd=conv3x3, executed_total=igemm + wg - reused, executed_igemm=igemm - reused)


def make_config(rank):
    from config import Config

    def synth_zca(seed=4321, dim=3072):
        """SURVEY §8d synthetic whitening constants: mean 0, seeded random orthogonal matrix (the real cifar10_zca_*.npy are not
        part of the reference repository).  Generated here: the timed path imports nothing from oracle/."""
        q, _ = np.linalg.qr(np.random.default_rng(seed).standard_normal((dim, dim)))
        return np.zeros(dim, np.float32), q.astype(np.float32)

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "cifar10"
        DATA_DIR = "/nonexistent"
        NUM_LABEL = 4000
        BATCH_SIZE_G = SIZES['B_G']
        BATCH_SIZE_L_C = SIZES['L_C']
        BATCH_SIZE_U_C = SIZES['U_C']
        BATCH_SIZE_L_D = SIZES['L_D']
        BATCH_SIZE_U_D = SIZES['U_D']
        BATCH_SIZE = SIZES['B_G']
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 32, 32, 3
        FAKE_G_LAMBDA = 0.3
        Z_DIM = 100
        NUM_CLASSES = 10
        LEARNING_RATE = 3e-4
        CLA_LEARNINIG_RATE = 3e-3
        EPOCHS = 1
        TRAIN_SIZE = 56000
        SUMMARY = False
        USE_HIP_GRAPH = None
        EXEC_MODE = 'auto'
        SEED = 0
        ZCA = synth_zca()
        RANK = rank

    return TempConfig()


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([int(i.get('num_threads', 1)) for i in threadpool_info() if i.get('user_api') == 'blas'] or [1])
    except Exception:
        return None


def cpu_baseline(warmup=1, timed=3):
    """the oracle as the CPU baseline ('port'; BASELINE.md §3 protocol, bounded): `warmup` untimed + `timed` timed iterations of the
    bench workload (one iteration = 100 nominal images), free-running from the same initial weights; value = 100 / median time."""
    from oracle import step_cifar10 as S
    st = S.new_state(S.init_params(0))
    zca = S.synth_zca()
    hyper = dict(lr=3e-4, cla_lr=3e-3, beta1=0.5, lambda_1=0.3, lambda_2=0.5)
    times = []
    for i in range(warmup + timed):
        batch, rnd = S.synth_batch(1 + i), S.synth_rnd(100 + i)
        t0 = time.perf_counter()
        S.train_step(st, batch, rnd, hyper, zca)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[warmup:]))
    threads = _blas_threads()
    return dict(value=round(SIZES['B_G'] / med, 3), unit="images/sec", cores=threads or os.cpu_count(), kind="port",
                host_logical_cpus=os.cpu_count(), blas_threads=threads, cpu_model=_cpu_model(),
                seconds_per_iteration=[round(t, 2) for t in times],
                sample="%d warm-up + %d timed iterations (median %.1f s) of the same CIFAR-10 bs=100 workload (100 nominal images, 1464 GFLOP "
                       "each), NumPy/BLAS fp32 oracle (TF1 unobtainable)" % (warmup, timed, med))


def measured_traffic():
    """HBM bytes of the dominant launch from the PMC passes stored under profiles/ (rocprofv3 cannot run inside this process).  The
    record names the kernel sources it was collected for (sha256 of the files in TRAFFIC_SOURCES): for any other source the
    figure is stale and `traffic` is null."""
    for name in ('r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json', 'r01_traffic.json'):
        tfile = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(tfile):
            continue
        rec = json.load(open(tfile))
        tj = rec.get('dominant_launch')
        if tj is None:
            return None, dict(incomplete='profiles/%s holds no record of the dominant launch: re-run tools/pmc_traffic.sh + tools/make_traffic_json.py' % name)
        want = rec.get('kernel_sources_sha256')
        csrc = os.path.join(PKG, 'csrc')
        import hashlib
        have = hashlib.sha256(b''.join(open(os.path.join(csrc, f), 'rb').read() for f in rec.get('kernel_source_files', TRAFFIC_SOURCES[:2]))).hexdigest()
        if want != have:
            return None, dict(stale='profiles/%s was collected for other kernel sources (%s...), this build has %s...: re-run '
                                    'tools/pmc_traffic.sh + tools/make_traffic_json.py' % (name, str(want)[:12], have[:12]))
        return tj['traffic_bytes_corrected'], dict(
            algorithmic_bytes_per_launch=tj['algorithmic_bytes'], kernel=tj['kernel'], kernel_sources_sha256=have,
            source='profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950-corrected)' % name)
    return None, None


def store_checksums(stores):
    """Per network two exact integers that identify the BITS of its flat parameter buffer: the sums of the low and of the high 16-bit
    halves of every float32 word (each < 2^40 for < 2^24 words: exact in int64 and in the float64 the exchange carries)."""
    import torch
    out = []
    for name in sorted(stores):
        bits = stores[name].p.detach().contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        out += [float((bits & 0xFFFF).sum().item()), float((bits >> 16).sum().item())]
    return out


def replicas_identical(stores, device):
    """every replica must hold bit-identical weights after the timed steps (identical initial weights + identical summed gradients +
    the same Adam arithmetic): min and max over the ranks of every checksum coincide."""
    from tg import dist as tgdist
    lo, hi = tgdist.minmax_over_ranks(store_checksums(stores), device)
    return lo == hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150)      # ~2.4 s timed at 16 ms/step
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--exec', dest='exec_mode', choices=('auto', 'plan', 'overlap', 'graph', 'eager'), default=os.environ.get('TG_EXEC_MODE', 'auto'),
                    help="auto: the trainer times plan and graph in the warm-up iterations and keeps the faster (default; TG_EXEC_MODE in the "
                         "environment overrides the default, so that a launcher that cannot pass flags can pin the mode); plan: the two-stream "
                         "launch sequence recorded once and re-issued natively (tg_plan_replay); overlap: the same sequence launched from Python; "
                         "graph: hipGraph replay; eager: one stream")
    ap.add_argument('--no-graph', action='store_true', help='alias of --exec eager')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--prof-iters', type=int, default=2)
    ap.add_argument('--soak-seconds', type=float, default=8.0,
                    help='untimed graph replays after the measurement so that a coarse GPU-utilisation sampler sees the run (0: none)')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no launcher above us: start the N ranks here.  This process has made no GPU call (it only counts devices) and stays the
        # parent; rank 0's JSON line goes straight to our stdout.
        from tg import launch
        raise SystemExit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    from tg import dist as tgdist
    from tg import lib
    from Training.Train_goodGAN import Train
    from Model.Good_GAN_cifar10 import Good_GAN_cifar10
    from Input_Pipeline.syntheticDataset import syntheticDataset

    world, rank, local = tgdist.env_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (or drop WORLD_SIZE and let bench.py spawn them)" % (args.gpus, world))
    if torch.cuda.device_count() <= local:
        raise SystemExit("rank %d needs HIP device %d, %d visible" % (rank, local, torch.cuda.device_count()))
    cfg = make_config(rank)
    cfg.EXEC_MODE = 'eager' if args.no_graph else args.exec_mode
    tr = Train(cfg, None, None)
    tr._build_train_graph(Good_GAN_cifar10)
    if tr.world != args.gpus or tgdist.world_size() != args.gpus or tgdist.rccl_ranks() != args.gpus:
        raise SystemExit("asked for %d replicas, the exchange has %d (communicator: %d)" % (args.gpus, tgdist.world_size(), tgdist.rccl_ranks()))
    tr.set_hyper(lambda_1=cfg.FAKE_G_LAMBDA, lambda_2=0.5)       # the late-training schedule: every loss term active
    cx = tr.cx
    exchange_ok_ranks = tgdist.self_test(cx.device)            # rank-id vector through the real exchange before anything is timed

    # synthetic batches, resident in HBM before the timed region (per-rank seed: SURVEY §8d)
    ds = syntheticDataset(None, cfg, cfg.NUM_LABEL, 'train', seed=1234 + rank)
    pool = []
    for _ in range(4):
        b = ds._next()
        pool.append({k: cx.from_numpy(v) for k, v in b.items()})

    def step(i):
        tr.feed(pool[i % len(pool)])
        tr.sample_latent()
        tr.train_iteration()

    # warm-up: the first call allocates eagerly, graph mode captures in the second; --exec auto decides within AUTO_ITERS iterations (three
    # alternating blocks per candidate: the first blocks of a fresh process measure page-ins, not the candidates) and runs a few more
    n_warm = max(args.warmup, 2) if cfg.EXEC_MODE != 'auto' else max(args.warmup, tr.AUTO_ITERS + 11)
    for i in range(n_warm):
        step(i)
    torch.cuda.synchronize()
    tgdist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_issue = time.perf_counter() - t0              # host time to ISSUE the timed steps (no device wait): the launch path's own cost
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    tgdist.barrier()
    dt = tgdist.max_over_ranks(dt_local, cx.device)
    dt_lo, dt_hi = tgdist.minmax_over_ranks([dt_local], cx.device)       # stragglers show as a gap between the fastest and the slowest rank
    losses = tr.losses()
    identical = replicas_identical(cx.stores, cx.device) if world > 1 else None
    # the launch path's own cost: host time to issue ONE iteration into an EMPTY queue (the figure above includes the time the host spends
    # blocked on a full queue once it runs ahead of the GPU — with native plan replay it does, and t_issue then reads as the step time)
    issue = []
    for i in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(i)
        issue.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    t_issue_free = float(np.median(issue))
    exposed_ms = None
    if world > 1:                                  # how much of the bucketed exchange the backward passes do not hide (untimed extra steps)
        n_x = min(20, max(args.steps, 1))
        tr.measure_exposed(True)
        for i in range(n_x):
            step(i)
        exposed_ms = tgdist.max_over_ranks(tr.exposed_ms() / n_x, cx.device)
        tr.measure_exposed(False)

    # ---- instrumented eager pass: per-kernel-class HIP-event timing on the launch stream
    fl = algorithmic_flops()
    args.prof_iters = max(args.prof_iters, 1)           # the roofline object needs at least one instrumented pass
    lib.call('tg_prof_reset')
    lib.call('tg_prof_enable', 1)
    timed_mode = cfg.EXEC_MODE
    cfg.EXEC_MODE = 'eager'                             # ONE stream: a kernel's HIP events must bracket that kernel alone
    for i in range(args.prof_iters):
        tr.feed(pool[i % len(pool)])
        tr.sample_latent()
        tr.train_iteration(use_graph=False)
    torch.cuda.synchronize()
    lib.call('tg_prof_enable', 0)
    cfg.EXEC_MODE = timed_mode
    import ctypes as C
    classes = {}
    for cls in range(lib.call('tg_prof_num_classes')):
        ms, n, f, b = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        lib.call('tg_prof_collect', cls, C.byref(ms), C.byref(n), C.byref(f), C.byref(b))
        name = lib.load().tg_prof_class_name(cls).decode()
        classes[name] = dict(ms_per_iter=ms.value / args.prof_iters, launches_per_iter=n.value / args.prof_iters,
                             executed_gflop_per_iter=f.value / args.prof_iters / 1e9, gbytes_per_iter=b.value / args.prof_iters / 1e9)
    dump = os.environ.get('TG_PROF_DUMP') or os.path.join('/tmp', 'tg_prof_%d.csv' % os.getpid())
    lib.call('tg_prof_dump', dump.encode())
    lib.call('tg_prof_reset')
    # the north-star kernel path: the seven 3x3 convolutions of the classifier (forward, input gradient, filter
    # gradient).  Their launches are recognised by geometry: 9 taps, classifier channel widths, stride 1.
    import csv
    import re
    conv_ms = {'igemm_f32': 0.0, 'wgrad_f32': 0.0}
    conv_n = {'igemm_f32': 0, 'wgrad_f32': 0}
    for row in csv.DictReader(open(dump)):
        m = re.match(r"M=(?:\d+x)?(\d+) N=(\d+) K=(\d+)x(\d+) in=(\d+)x\d+ s=(\d+)", row['desc'] or '')
        if not m or row['class'] not in conv_ms:
            continue
        mm, n, taps, ld, hin, st = (int(v) for v in m.groups())
        conv1_1 = taps == 1 and ld == 32 and n == 128 and hin == 32        # first conv runs as a 1x1 product on 3x3 patches
        if conv1_1 or (taps == 9 and st == 1 and ld in (128, 256, 512) and n in (128, 256, 512) and hin in (32, 16, 8, 6)):
            conv_ms[row['class']] += float(row['ms']) / args.prof_iters
            conv_n[row['class']] += 1
    # the dominant launch on its own: the 250-image conv1_2 / conv1_3 launch of conv3x3_pipe_kernel<32> (75.5 GFLOP each) — one row of
    # profiles/rNN_kernel_stats.csv reproduces this figure
    dom = [float(r['ms']) for r in csv.DictReader(open(dump))
           if r['class'] == 'igemm_f32' and (r['desc'] or '').startswith("M=1x%d N=128 K=9x128 in=32x32" % (1024 * (SIZES['L_C'] + 2 * SIZES['U_C'] + SIZES['B_G'])))]
    if not os.environ.get('TG_PROF_DUMP'):
        os.remove(dump)
    s_ = SIZES
    n_c = s_['L_C'] + 2 * s_['U_C'] + s_['B_G']
    n_cd = s_['U_C'] + s_['U_D']
    mf = lambda h, ci, co: 2.0 * h * h * 9 * ci * co
    c3 = [mf(32, 3, 128), mf(32, 128, 128), mf(32, 128, 128), mf(16, 128, 256), mf(16, 256, 256), mf(16, 256, 256), mf(6, 256, 512)]
    conv_fl_ig = (n_c + n_cd) * sum(c3) + n_c * sum(c3[1:])          # forward (C- and D-update) + input gradients
    conv_fl_wg = n_c * sum(c3)
    conv_tf_ig = conv_fl_ig / (conv_ms['igemm_f32'] * 1e-3) / 1e12
    conv_tf_wg = conv_fl_wg / (conv_ms['wgrad_f32'] * 1e-3) / 1e12
    conv_tf = (conv_fl_ig + conv_fl_wg) / ((conv_ms['igemm_f32'] + conv_ms['wgrad_f32']) * 1e-3) / 1e12
    ig = classes['igemm_f32']
    achieved = fl['executed_igemm'] / (ig['ms_per_iter'] * 1e-3) / 1e12
    n_conv_launches = (conv_n['igemm_f32'] + conv_n['wgrad_f32']) / args.prof_iters
    traffic, traffic_detail = measured_traffic()
    dom_gflop = (s_['L_C'] + 2 * s_['U_C'] + s_['B_G']) * c3[1] / 1e9
    dominant = None
    if dom:
        dom_ms = sum(dom) / len(dom)
        dominant = dict(kernel="conv3x3_pipe_kernel<32, COLSUM, fp32> on one 250-image 32x32 128->128 launch (conv1_2 / conv1_3 forward and input gradient)",
                        launches_sampled=len(dom), avg_launch_ms=round(dom_ms, 5), algorithmic_gflop_per_launch=round(dom_gflop, 2),
                        achieved=round(dom_gflop / dom_ms, 2), frac=round(dom_gflop / dom_ms / PEAK_FP32_MFMA_TFLOPS, 4))
    roofline = dict(bound="mfma", kernel="classifier 3x3 conv path: conv3x3_pipe_kernel / igemm_f32_kernel (fwd + input grad) + wgrad3x3_kernel / wgrad_f32_kernel (filter grad)",
                    achieved=round(conv_tf, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(conv_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                    traffic=traffic, traffic_detail=traffic_detail, launches_per_step=n_conv_launches,
                    avg_launch_ms=round((conv_ms['igemm_f32'] + conv_ms['wgrad_f32']) / max(n_conv_launches, 1), 5),
                    algorithmic_gflop_per_step=round((conv_fl_ig + conv_fl_wg) / 1e9, 1),
                    conv3x3_igemm=dict(achieved=round(conv_tf_ig, 2), ms_per_step=round(conv_ms['igemm_f32'], 3)),
                    conv3x3_wgrad=dict(achieved=round(conv_tf_wg, 2), ms_per_step=round(conv_ms['wgrad_f32'], 3)),
                    all_igemm_launches=dict(achieved=round(achieved, 2), frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                                            launches_per_step=ig['launches_per_iter'], algorithmic_gflop_per_step=round(fl['executed_igemm'] / 1e9, 1)),
                    all_wgrad_launches=dict(achieved=round(fl['wgrad'] / (classes['wgrad_f32']['ms_per_iter'] * 1e-3) / 1e12, 2),
                                            algorithmic_gflop_per_step=round(fl['wgrad'] / 1e9, 1)),
                    dominant_launch=dominant,
                    class_ms_per_step={k: round(v['ms_per_iter'], 3) for k, v in classes.items()})
    if args.soak_seconds > 0:                      # untimed: keeps the GPU visibly busy for a sampler that looks every few seconds
        t_end = time.perf_counter() + args.soak_seconds
        while time.perf_counter() < t_end:
            for i in range(20):
                step(i)
            torch.cuda.synchronize()

    if rank == 0:
        out = {
            "metric": "Triple-GAN train images/sec (G+C+D step) CIFAR-10 32x32 bs=100",
            "value": round(args.steps * SIZES['B_G'] * world / dt, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "rccl_ranks": tgdist.rccl_ranks(),
            "dist_backend": tgdist.backend_name(),
            "exchange_self_test_ranks": exchange_ok_ranks,
            "replicas_identical": identical,
            "exchange_exposed_ms": None if exposed_ms is None else round(exposed_ms, 4),
            "steps": args.steps,
            "warmup": args.warmup, "warmup_executed": n_warm,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_step_rank_min": round(dt_lo[0] / args.steps * 1e3, 4), "ms_per_step_rank_max": round(dt_hi[0] / args.steps * 1e3, 4),
            "host_issue_ms_per_step": round(t_issue_free * 1e3, 4), "host_issue_ms_per_step_queue_full": round(t_issue / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "CIFAR-10 32x32x3, 4000 labelled, bs=100 fp32 (B_G/L_C/U_C/L_D/U_D=100/50/50/20/80), "
                                   "Good_GAN_cifar10 D+G+C step", "global_batch": SIZES['B_G'] * world, "parallelism": "dp%d" % world,
                       "exec_mode": cfg.EXEC_MODE, "exec_mode_chosen": tr.exec_mode_chosen()[0] if cfg.EXEC_MODE == 'auto' else cfg.EXEC_MODE,
                       "exec_mode_timings_ms": {k: round(v * 1e3, 3) for k, v in tr.exec_mode_chosen()[1].items()},
                       "exec_mode_blocks_ms": {k: [round(x * 1e3, 2) for x in v] for k, v in getattr(tr, '_auto', {}).get('full', {}).get('t', {}).items()},
                       "hip_graph": (tr.exec_mode_chosen()[0] if cfg.EXEC_MODE == 'auto' else cfg.EXEC_MODE) == 'graph', "algorithmic_gflop_per_step": round(fl['total'] / 1e9, 1),
                       "executed_gflop_per_step": round(fl['executed_total'] / 1e9, 1),
                       "step_tflops": round(fl['executed_total'] / (dt / args.steps) / 1e12, 2),
                       "step_tflops_algorithmic": round(fl['total'] / (dt / args.steps) / 1e12, 2), "losses_d_g_c": [round(v, 4) for v in losses]},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    tgdist.barrier()
    tgdist.shutdown()


if __name__ == "__main__":
    main()
