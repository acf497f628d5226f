"""The classifier's fused conv + column-sum launches (tg_igemm_colsum_f32) under each tile candidate vs the model's pick."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom

lib.load()


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


for name, segs, hw, ci, co, pad in (("conv1_2 C-phase", (50, 100, 100), 32, 128, 128, 'SAME'), ("conv1_2 D-phase", (50, 80), 32, 128, 128, 'SAME'),
                                    ("conv2_1 C-phase", (50, 100, 100), 16, 128, 256, 'SAME'), ("conv2_2 C-phase", (50, 100, 100), 16, 256, 256, 'SAME'),
                                    ("conv2_2 D-phase", (50, 80), 16, 256, 256, 'SAME'), ("conv3 C-phase", (50, 100, 100), 8, 256, 512, 'VALID')):
    n = sum(segs)
    d = geom.conv_fwd(n, hw, hw, ci, co, 3, 1, pad)
    ho = d.h_out
    x = torch.randn(n, hw, hw, ci, device='cuda').clamp_(min=-0.2)
    w = torch.randn(co, 9, ci, device='cuda') * 0.05
    y = torch.empty(n, ho, ho, co, device='cuda')
    sums = torch.zeros(len(segs) * co, dtype=torch.float64, device='cuda')
    seg = (C.c_int32 * len(segs))(*[s * ho * ho for s in segs])
    st = lib.cur_stream()
    call = lambda: lib.call("tg_igemm_colsum_f32", d, lib.ptr(x), lib.ptr(w), lib.ptr(y), seg, len(segs), lib.ptr(sums), 0, None, 0, st)
    fl = 2.0 * n * ho * ho * co * 9 * ci
    os.environ.pop("TG_IGEMM_TILE", None)
    timeit(call)
    res = {}
    for t in ("128,128", "64,128", "64,64", "128,64"):
        os.environ["TG_IGEMM_TILE"] = t
        try:
            res[t] = timeit(call)
        except lib.TgError:
            pass
    os.environ.pop("TG_IGEMM_TILE", None)
    tm = timeit(call)
    print("%-16s model %.4f ms %6.1f TF | " % (name, tm, fl / tm / 1e9) + "  ".join("%s %.4f" % kv for kv in res.items()), flush=True)
