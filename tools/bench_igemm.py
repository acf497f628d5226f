"""Micro-benchmark of the MFMA implicit-GEMM kernels at the classifier's layer shapes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom

lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
PREC = sys.argv[2] if len(sys.argv) > 2 else 'f32'      # 'f32' | 'bf16'
LAYERS = [("conv1_2 128->128 @32", 32, 128, 128, 3, 'SAME'), ("conv2_1 128->256 @16", 16, 128, 256, 3, 'SAME'),
          ("conv2_2 256->256 @16", 16, 256, 256, 3, 'SAME'), ("conv3 256->512 @8 VALID", 8, 256, 512, 3, 'VALID'),
          ("NiN1 512->256 @6", 6, 512, 256, 1, 'SAME')]


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


for name, hw, ci, co, k, pad in LAYERS:
    x = torch.randn(N, hw, hw, ci, device='cuda')
    w = torch.randn(co, k * k, ci, device='cuda') * 0.05
    d = geom.conv_fwd(N, hw, hw, ci, co, k, 1, pad)
    ho = d.h_out
    y = torch.empty(N, ho, ho, co, device='cuda')
    st = lib.cur_stream()
    ms = timeit(lambda: lib.call_igemm("tg_igemm_" + PREC, d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), st))
    fl = 2.0 * N * ho * ho * co * k * k * ci
    print("fwd   %-26s %8.3f ms  %7.1f TFLOP/s" % (name, ms, fl / ms / 1e9))
    if os.environ.get('FWD_ONLY'):
        continue
    dy = torch.randn(N, ho, ho, co, device='cuda')
    dw = geom.conv_wgrad(N, hw, hw, ci, co, k, 1, pad)
    tiles = k * k * (ci // 128) * (co // 128)
    nsplit = max(1, min(512 // tiles, (N * ho * ho) // 512))
    slab = torch.empty(nsplit, k * k, ci, co, device='cuda')
    ms = timeit(lambda: lib.call("tg_wgrad_" + PREC, dw, lib.ptr(x), lib.ptr(dy), lib.ptr(slab), nsplit, st))
    print("wgrad %-26s %8.3f ms  %7.1f TFLOP/s  (split %d)" % (name, ms, fl / ms / 1e9, nsplit))
