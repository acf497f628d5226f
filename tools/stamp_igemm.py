"""Diagnostic: per-phase cycle shares of igemm_f32_kernel's K loop (libtg_stamp.so built with -DTG_STAMP)."""
import ctypes as C, os, sys
os.environ['TG_LIB'] = 'libtg_stamp.so'
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom
L = lib.load()
N = 250
for name, hw, ci, co in (("conv1_2", 32, 128, 128), ("conv2_2", 16, 256, 256)):
    x = torch.randn(N, hw, hw, ci, device='cuda'); w = torch.randn(co, 9, ci, device='cuda') * 0.05
    d = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME'); y = torch.empty(N, hw, hw, co, device='cuda')
    for _ in range(5):
        lib.call("tg_igemm_f32", d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, lib.cur_stream())
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    L.tg_debug_read_stamps.argtypes = [C.POINTER(C.c_uint64)]
    L.tg_debug_read_stamps(buf)
    for s in range(5):
        ld, mf, st, ba, tot, nk, pro, epi = [buf[s * 8 + i] for i in range(8)]
        if nk:
            print("%s slot%d nk=%d per-iter cycles: load-issue %.0f  mfma %.0f  store %.0f  barrier %.0f  | loop total/iter %.0f | loop %d prologue %d epilogue %d" %
                  (name, s, nk, ld / nk, mf / nk, st / nk, ba / nk, tot / nk, tot, pro, epi))
