"""Diagnostic: per-phase cycle shares of igemm_f32_kernel's K loop on the SMALL layers (discriminator convs, generator's last stages);
libtg_stamp.so (make -C csrc libtg_stamp.so)."""
import ctypes as C, os, sys
os.environ['TG_LIB'] = 'libtg_stamp.so'
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom
L = lib.load()
L.tg_debug_read_stamps.argtypes = [C.POINTER(C.c_uint64)]
for name, N, hw, ci, co in (("d_conv0 32ch", 100, 32, 32, 32), ("d_conv2 64->64 @16", 100, 16, 64, 64), ("d_conv4 96->128 @8", 250, 8, 96, 128), ("d_conv5 160->128 @8 n=100 (200 units)", 100, 8, 160, 128),
                            ("d_conv5 160->128 @8 n=250 (500 units)", 250, 8, 160, 128), ("conv3-like 256->512 @8 n=250 SAME via generic (2000 units)", 250, 8, 256, 512),
                            ("c_conv1_2 (reference)", 250, 32, 128, 128)):
    x = torch.randn(N, hw, hw, ci, device='cuda'); w = torch.randn(co, 9, ci, device='cuda') * 0.05
    d = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME'); y = torch.empty(N, hw, hw, co, device='cuda')
    for _ in range(5):
        lib.call("tg_igemm_f32", d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, lib.cur_stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.call("tg_igemm_f32", d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, lib.cur_stream())
    b.record(); torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    L.tg_debug_read_stamps(buf)
    print("%s: %.1f us per launch" % (name, a.elapsed_time(b) / 20 * 1000))
    for s in range(5):
        ld, mf, st, ba, tot, nk, pro, epi = [buf[s * 8 + i] for i in range(8)]
        if nk:
            print("   slot%d nk=%d per-iter cycles: load-issue %.0f  mfma %.0f  store %.0f  barrier %.0f  | loop total/iter %.0f | loop %d prologue %d epilogue %d" %
                  (s, nk, ld / nk, mf / nk, st / nk, ba / nk, tot / nk, tot, pro, epi))
