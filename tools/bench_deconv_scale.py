"""How the generator's transposed-conv forward launches scale with the batch (quantisation / tail effects vs per-tile efficiency)."""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom

lib.load()
LAYERS = [("dconv0 4x4x544->256", 4, 544, 256), ("dconv1 8x8x288->128", 8, 288, 128)]


def timeit(fn, iters=30):
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


for name, hw, ci, co in LAYERS:
    for n in (100, 200, 400, 1600):
        x = torch.randn(n, hw, hw, ci, device='cuda')
        w = torch.randn(25, co, ci, device='cuda') * 0.05
        y = torch.empty(n, 2 * hw, 2 * hw, co, device='cuda')
        st = lib.cur_stream()
        fl = 2.0 * n * hw * hw * 25 * ci * co
        dds = lib.desc_array(geom.deconv_fwd(n, hw, hw, ci, co))
        ms = timeit(lambda: lib.call("tg_igemm_multi_f32", C.cast(dds, C.c_void_p), len(dds), lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, st))
        # the four parities as four separate launches (no imbalance inside a launch, but four tails)
        singles = [lib.desc_array([d]) for d in geom.deconv_fwd(n, hw, hw, ci, co)]
        def four():
            for d1 in singles:
                lib.call("tg_igemm_multi_f32", C.cast(d1, C.c_void_p), 1, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, st)
        ms4 = timeit(four)
        print("%-22s n=%5d  one launch %7.3f ms %6.1f TFLOP/s   four launches %7.3f ms %6.1f TFLOP/s" % (name, n, ms, fl / ms / 1e9, ms4, fl / ms4 / 1e9), flush=True)
