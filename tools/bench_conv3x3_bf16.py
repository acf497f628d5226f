#!/usr/bin/env python3
"""The classifier's 3x3 layers on the halo kernels (csrc/conv3x3_bf16.hip, csrc/wgrad3x3.hip), N = 250 images (TG_BENCH_N), standard-normal
operands: TFLOP/s of tg_igemm_bf16 / tg_igemm_colsum_bf16 / tg_igemm_f32 (forward) and tg_wgrad_bf16 / tg_wgrad_f32 (filter gradient, pixel
split from tg_wgrad_splits[_bf16]).  TG_BENCH_ONLY=<bf16|bf16_colsum|f32|wgrad_bf16|wgrad_f32> runs one of them; the generic kernels for
comparison: TG_NO_CONV3X3_BF16=1 / TG_NO_CONV3X3_F32=1 / TG_NO_WGRAD3X3=1.  One JSON
line per layer; bf16 fractions against 2 500 TFLOP/s dense bf16 (MI355X_MICROARCH.md)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
import torch  # noqa: E402
from tg import geom, lib  # noqa: E402

PEAK_BF16 = 2500.0
N = int(os.environ.get('TG_BENCH_N', '250'))
LAYERS = [('conv1_2', 32, 128, 128), ('conv2_1', 16, 128, 256), ('conv2_2', 16, 256, 256)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


lib.load()
st = lib.cur_stream()
for name, hw, ci, co in LAYERS:
    x = torch.randn(N, hw, hw, ci, device='cuda')
    w = torch.randn(co, 9, ci, device='cuda') * 0.05
    y = torch.empty(N, hw, hw, co, device='cuda')
    sums = torch.zeros(2 * co, dtype=torch.float32, device='cuda')
    d = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME', act='lrelu')
    dc = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME')
    seg = (C.c_int32 * 1)(N * hw * hw)
    gf = 2.0 * N * hw * hw * 9 * ci * co / 1e9
    out = dict(layer=name, gflop=round(gf, 1))
    only = os.environ.get('TG_BENCH_ONLY')
    dw = geom.conv_wgrad(N, hw, hw, ci, co, 3, 1, 'SAME')
    ns, ns16 = geom.wgrad_splits(dw), geom.wgrad_splits(dw, True)
    slab = torch.empty(max(ns, ns16) * 9 * ci * co, device='cuda')
    wpk = torch.empty(max(lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, 1), 16) // 4, device='cuda')   # caller-owned scratch of the bf16 3x3 kernel
    for tag, fn in (('wgrad_bf16', lambda: lib.call('tg_wgrad_bf16', dw, lib.ptr(x), lib.ptr(y), lib.ptr(slab), ns16, st)),
                    ('wgrad_f32', lambda: lib.call('tg_wgrad_f32', dw, lib.ptr(x), lib.ptr(y), lib.ptr(slab), ns, st)),
                    ('bf16', lambda: lib.call('tg_igemm_bf16', d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), lib.ptr(wpk), wpk.numel() * 4, st)),
                    ('bf16_colsum', lambda: lib.call('tg_igemm_colsum_bf16', dc, lib.ptr(x), lib.ptr(w), lib.ptr(y), seg, 1, lib.ptr(sums), 0, lib.ptr(wpk), wpk.numel() * 4, st)),
                    ('f32', lambda: lib.call('tg_igemm_f32', d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, st))):
        if only and tag != only:
            continue
        ms = timeit(fn)
        out[tag] = dict(ms=round(ms, 4), tflops=round(gf / ms, 1), frac_of_bf16_peak=round(gf / ms / PEAK_BF16, 4) if 'f32' not in tag else None)
    print(json.dumps(out), flush=True)
