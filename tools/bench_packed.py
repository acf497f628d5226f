#!/usr/bin/env python3
"""The discriminator's first layer (3x3, 13 -> 32 channels, 32x32 images) on the K-packed kernels of csrc/packed_conv.hip against the generic
implicit GEMM / filter gradient it replaces: us per launch at the step's three batch sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
import torch  # noqa: E402
from tg import geom, lib  # noqa: E402

lib.load()
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


h = w = 32
cin, cout, nlab = 13, 32, 10
ci_p, ld = 32, 64
for n in (250, 100, 50):
    x = torch.randn(n, h, w, ci_p, device='cuda')
    wt = torch.randn(3, 3, cin, cout, device='cuda') * 0.1
    bias = torch.randn(cout, device='cuda')
    lab = torch.rand(n, nlab, device='cuda')
    y = torch.empty(n, h, w, ld, device='cuda')
    dy = torch.randn(n, h, w, 32, device='cuda')
    dw = torch.empty(3, 3, cin, cout, device='cuda')
    ws = torch.empty(lib.call('tg_conv3x3_packed_wgrad_workspace_bytes', n, h, w, cin, cout) // 4, device='cuda')
    t_f = timeit(lambda: lib.call('tg_conv3x3_packed_fwd_f32', lib.ptr(x), ci_p, cin, lib.ptr(wt), lib.ptr(bias), lib.ACT['lrelu'], 0.2, lib.ptr(lab), nlab,
                                  lib.ptr(y), ld, n, h, w, cout, st))
    t_w = timeit(lambda: lib.call('tg_conv3x3_packed_wgrad_f32', lib.ptr(x), ci_p, cin, lib.ptr(dy), 32, n, h, w, cout, lib.ptr(ws), lib.ptr(dw), st))
    # the generic kernels on the same layer
    w_oti = torch.randn(32 * 9 * ci_p, device='cuda') * 0.1
    d = geom.conv_fwd(n, h, w, ci_p, 32, 3, 1, 'SAME', ld_out=ld, n_store=cout, act='lrelu')
    t_g = timeit(lambda: lib.call_igemm('tg_igemm_labels_f32', d, lib.ptr(x), lib.ptr(w_oti), lib.ptr(bias), lib.ptr(lab), nlab, lib.ptr(y), st))
    dd = geom.conv_wgrad(n, h, w, ci_p, 32, 3, 1, 'SAME')
    ns = geom.wgrad_splits(dd, False)
    slab = torch.empty(geom.wgrad_slab_floats(dd, ns), device='cuda')
    dst = torch.empty(9 * cin * cout, device='cuda')
    def generic_wgrad():
        lib.call('tg_wgrad_f32', dd, lib.ptr(x), lib.ptr(dy), lib.ptr(slab), ns, st)
        lib.call('tg_slab_reduce_f32', lib.ptr(slab), ns, 9, ci_p, 32, cin, cout, lib.ptr(dst), st)
    t_gw = timeit(generic_wgrad)
    print("n=%3d  forward: packed %.1f us, generic %.1f us   filter gradient (+ reduce): packed %.1f us, generic %.1f us" % (n, t_f, t_g, t_w, t_gw))
