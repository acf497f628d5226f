#!/usr/bin/env python3
"""The generator's image layer backward on the vector ALUs (csrc/narrow.hip) at the CIFAR-10 step's shape: 100 images, 16x16x138 -> 32x32x3.
Prints us per launch of tg_deconv5x5s2_narrow_dgrad_f32 and tg_deconv5x5s2_narrow_wgrad_f32 (TG_LIB selects an A/B build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
import torch  # noqa: E402
from tg import geom, lib  # noqa: E402

lib.load()
n, h, w, cin, cout = 100, 16, 16, 138, 3
ci_p, co_p = geom.pad32(cin), geom.pad32(cout)
x = torch.randn(n, h, w, ci_p, device='cuda')
wt = torch.randn(5, 5, cout, cin, device='cuda') * 0.1
dy = torch.randn(n, 2 * h, 2 * w, co_p, device='cuda')
dx = torch.empty(n, h, w, ci_p, device='cuda')
dw = torch.empty(25, cout, cin, device='cuda')
ws = torch.empty(lib.call('tg_deconv5x5s2_narrow_wgrad_workspace_bytes', n, h, w, cout, ci_p) // 4, device='cuda')
st = lib.cur_stream()


def timeit(fn, iters=300):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


t_d = timeit(lambda: lib.call('tg_deconv5x5s2_narrow_dgrad_f32', lib.ptr(dy), co_p, lib.ptr(wt), None, n, h, w, cout, cin, ci_p, lib.ptr(dx), ci_p, st))
t_w = timeit(lambda: lib.call('tg_deconv5x5s2_narrow_wgrad_f32', lib.ptr(dy), co_p, lib.ptr(x), ci_p, n, h, w, cout, cin, ci_p, lib.ptr(ws), lib.ptr(dw), st))
fl = 2.0 * n * h * w * cin * 25 * cout
print("narrow dgrad %.1f us (%.2f TFLOP/s)   narrow wgrad + reduce %.1f us (%.2f TFLOP/s)   [%s]" % (t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6, os.path.basename(lib.LIB_PATH)))
