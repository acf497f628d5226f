"""Micro-benchmark of the generator's transposed-conv forward launches under forced igemm tiles (TG_IGEMM_TILE)."""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom

lib.load()
LAYERS = [("dconv0 4x4x544->256", 100, 4, 544, 256), ("dconv1 8x8x288->128", 100, 8, 288, 128)]


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


for name, n, hw, ci, co in LAYERS:
    x = torch.randn(n, hw, hw, ci, device='cuda')
    w = torch.randn(25, co, ci, device='cuda') * 0.05
    y = torch.empty(n, 2 * hw, 2 * hw, co, device='cuda')
    dds = lib.desc_array(geom.deconv_fwd(n, hw, hw, ci, co))
    st = lib.cur_stream()
    fl = 2.0 * n * hw * hw * 25 * ci * co
    for tile in (None, "128,128", "64,128", "64,64", "128,64", "32,128", "128,32"):
        if tile is None:
            os.environ.pop("TG_IGEMM_TILE", None)
        else:
            os.environ["TG_IGEMM_TILE"] = tile
        try:
            ms = timeit(lambda: lib.call("tg_igemm_multi_f32", C.cast(dds, C.c_void_p), len(dds), lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, st))
            print("%-22s tile %-8s %7.3f ms %6.1f TFLOP/s" % (name, tile or "model", ms, fl / ms / 1e9))
        except lib.TgError as e:
            print(name, tile, "n/a")
