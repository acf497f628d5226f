#!/usr/bin/env python3
"""Every launch shape of the CIFAR-10 step that the GENERIC MFMA kernels serve (igemm_f32_kernel / wgrad_f32_kernel: discriminator,
generator, conv3 / NiN, ZCA, dense — profiles/rNN_launches.csv), timed one by one through the C ABI with the launch's own scratch.

    python tools/bench_step_shapes.py [f32|bf16] [csv path]          TG_LIB=libtg_<tag>.so selects an A/B build of the library

Prints per shape: ms, executed GFLOP, TFLOP/s, launches per step, and the per-step total — the A/B harness of kernel changes (the
in-step times differ by cache state and by what runs beside them; the ORDER between builds carries over)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom  # noqa: E402

lib.load()
PREC = sys.argv[1] if len(sys.argv) > 1 else 'f32'
CSV = sys.argv[2] if len(sys.argv) > 2 else None
p32 = geom.pad32


ITERS = int(os.environ.get('TG_SHAPES_ITERS', '30'))          # PMC passes (tools/pmc_step_shapes.sh) use few


def timeit(fn, iters=None):
    iters = ITERS if iters is None else iters
    for _ in range(min(5, iters)):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def run_igemm(descs, n_in, n_w, n_out):
    x = torch.randn(n_in, device='cuda')
    w = torch.randn(n_w, device='cuda') * 0.05
    y = torch.empty(n_out, device='cuda')
    st = lib.cur_stream()
    if isinstance(descs, list):
        arr = lib.desc_array(descs)
        name, args = 'tg_igemm_multi_' + PREC, (arr, len(descs), lib.ptr(x), lib.ptr(w), None, lib.ptr(y), st)
        d0 = descs
    else:
        name, args = 'tg_igemm_' + PREC, (descs, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), st)
        d0 = [descs]
    need = lib.igemm_workspace_bytes(name, args)
    ws = torch.empty(max(need // 4, 4), device='cuda')
    full = args[:-1] + (lib.ptr(ws) if need else None, need, st)
    ms = timeit(lambda: lib.call(name, *full))
    M = d0[0].n_img * d0[0].h_v * d0[0].w_v
    fl = sum(2.0 * M * d.c_out * d.n_taps * d.ld_in for d in d0)
    return ms, fl


def run_wgrad(d, n_in, n_dy):
    x = torch.randn(n_in, device='cuda')
    dy = torch.randn(n_dy, device='cuda')
    ns = geom.wgrad_splits(d, PREC == 'bf16')
    slab = torch.empty(geom.wgrad_slab_floats(d, ns), device='cuda')
    st = lib.cur_stream()
    ms = timeit(lambda: lib.call('tg_wgrad_' + PREC, d, lib.ptr(x), lib.ptr(dy), lib.ptr(slab), ns, st))
    M = d.n_img * d.h_v * d.w_v
    return ms, 2.0 * M * d.c_out * d.n_taps * d.ld_in


ROWS = []


def conv(tag, n, h, ci, co, k, s, pad, per_step, fwd=True, dgrad=False, wgrad=False, ld_out=None):
    ci_p, co_p = p32(ci), p32(co)
    ho = geom.conv_fwd(n, h, h, ci_p, co_p, k, s, pad).h_out
    if fwd:
        d = geom.conv_fwd(n, h, h, ci_p, co_p, k, s, pad, ld_out=ld_out, act='lrelu')
        ROWS.append((tag + ' fwd', per_step['fwd'], *run_igemm(d, n * h * h * ci_p, co_p * k * k * ci_p, n * ho * ho * (ld_out or co_p))))
    if dgrad:
        dl = geom.conv_dgrad(n, h, h, ci_p, co_p, k, s, pad)
        ROWS.append((tag + ' dgrad', per_step['dgrad'], *run_igemm(dl if len(dl) > 1 else dl[0], n * ho * ho * co_p, k * k * ci_p * co_p, n * h * h * ci_p)))
    if wgrad:
        d = geom.conv_wgrad(n, h, h, ci_p, co_p, k, s, pad)
        ROWS.append((tag + ' wgrad', per_step['wgrad'], *run_wgrad(d, n * h * h * ci_p, n * ho * ho * co_p)))


def deconv(tag, n, h, ci, co, per_step, dgrad=True, wgrad=True):
    ci_p, co_p = p32(ci), p32(co)
    dl = geom.deconv_fwd(n, h, h, ci_p, co_p, act='relu')
    ROWS.append((tag + ' fwd', per_step['fwd'], *run_igemm(dl, n * h * h * ci_p, 25 * co_p * ci_p, n * 4 * h * h * co_p)))
    if dgrad:
        d = geom.deconv_dgrad(n, h, h, ci_p, co_p)
        ROWS.append((tag + ' dgrad', per_step['dgrad'], *run_igemm(d, n * 4 * h * h * co_p, 25 * ci_p * co_p, n * h * h * ci_p)))
    if wgrad:
        d = geom.deconv_wgrad(n, h, h, co_p, ci_p)
        ROWS.append((tag + ' wgrad', per_step['wgrad'], *run_wgrad(d, n * 4 * h * h * co_p, n * h * h * ci_p)))


# ---- discriminator (Model/Good_GAN_cifar10.py:60-99): (cin incl. label channels, cout, input size, stride)
D_LAYERS = [(13, 32, 32, 1), (42, 32, 32, 2), (42, 64, 16, 1), (74, 64, 16, 2), (74, 128, 8, 1), (138, 128, 8, 1)]
for n, what in ((250, 'D-update'), (100, 'G-update'), (50, 'C-update')):
    for i, (ci, co, h, s) in enumerate(D_LAYERS):
        first = i == 0
        # forward in all three updates; input gradient: D-update (not for the first layer), G-update (all: the image needs it), C-update none
        # (the labels are arg-max one-hots); filter gradient: D-update only
        conv('D%d %dx%d->%d @%d s%d n=%d' % (i + 1, 3, ci, co, h, s, n), n, h, ci, co, 3, s, 'SAME', dict(fwd=1, dgrad=1, wgrad=1),
             fwd=True, dgrad=(what == 'D-update' and not first) or what == 'G-update', wgrad=what == 'D-update')
# ---- generator (:33-58): dense 110 -> 8192, deconv 522 -> 256 @4, 266 -> 128 @8 (the image layer runs merged / on the vector ALUs)
for tag, n_fwd in (('G', 2),):
    d = geom.dense_fwd(100, 128, 8192, act='relu')
    ROWS.append(('G dense 110->8192 fwd', 2, *run_igemm(d, 100 * 128, 8192 * 128, 100 * 8192)))
    deconv('G deconv 522->256 @4 n=100', 100, 4, 522, 256, dict(fwd=2, dgrad=1, wgrad=1))
    deconv('G deconv 266->128 @8 n=100', 100, 8, 266, 128, dict(fwd=2, dgrad=1, wgrad=1))
# ---- classifier tail (:145-172): conv3 VALID 8 -> 6, NiN 512 -> 256 -> 128; 130 images in the D-update, 250 in the C-update
for n, tr in ((130, False), (250, True)):
    conv('C conv3 256->512 VALID @8 n=%d' % n, n, 8, 256, 512, 3, 1, 'VALID', dict(fwd=1, dgrad=1, wgrad=1), dgrad=tr, wgrad=tr)
    conv('C NiN1 512->256 @6 n=%d' % n, n, 6, 512, 256, 1, 1, 'SAME', dict(fwd=1, dgrad=1, wgrad=1), dgrad=tr, wgrad=tr)
    conv('C NiN2 256->128 @6 n=%d' % n, n, 6, 256, 128, 1, 1, 'SAME', dict(fwd=1, dgrad=1, wgrad=1), dgrad=tr, wgrad=tr)
    # conv1_1 as a 1x1 product over the 27 (-> 32) im2col channels
    conv('C conv1_1 27->128 @32 n=%d' % n, n, 32, 27, 128, 1, 1, 'SAME', dict(fwd=1, dgrad=0, wgrad=1), wgrad=tr)

tot = {}
print("%-44s %5s %9s %9s %8s" % ("shape", "x", "ms", "GFLOP", "TFLOP/s"))
for tag, mult, ms, fl in ROWS:
    kind = 'wgrad' if tag.endswith('wgrad') else 'igemm'
    tot[kind] = tot.get(kind, 0.0) + mult * ms
    print("%-44s %5d %9.4f %9.2f %8.1f" % (tag, mult, ms, fl / 1e9, fl / ms / 1e9))
print("per step: igemm %.3f ms, wgrad %.3f ms (library %s)" % (tot.get('igemm', 0), tot.get('wgrad', 0), os.path.basename(lib.LIB_PATH)))
if CSV:
    with open(CSV, 'w') as f:
        f.write("shape,per_step,ms,gflop\n")
        for tag, mult, ms, fl in ROWS:
            f.write("%s,%d,%.5f,%.3f\n" % (tag, mult, ms, fl / 1e9))
