"""Diagnostic: cycles per phase of conv3x3_pipe_kernel (libtg_stamp.so, -DTG_STAMP): prologue load issue, prologue (until the first barrier),
K loop, epilogue — wave 0 of the first 64 workgroups, median."""
import ctypes as C, os, sys
os.environ['TG_LIB'] = 'libtg_stamp.so'
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom
L = lib.load()
N = 250
PREC = sys.argv[1] if len(sys.argv) > 1 else 'bf16'          # f32: the exact-fp32 form of the same kernel (32-channel chunks)
for name, hw, ci, co in (("conv1_2", 32, 128, 128), ("conv2_2", 16, 256, 256)):
    x = torch.randn(N, hw, hw, ci, device='cuda'); w = torch.randn(co, 9, ci, device='cuda') * 0.05
    d = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME'); y = torch.empty(N, hw, hw, co, device='cuda')
    wpk = torch.empty(max(lib.call('tg_igemm_workspace_bytes', C.byref(d), 1, None, 0, 1 if PREC == 'bf16' else 0), 16) // 4, device='cuda')
    for _ in range(5):
        lib.call("tg_igemm_" + PREC, d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), lib.ptr(wpk), wpk.numel() * 4, lib.cur_stream())
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 512)()
    L.tg_debug_read_conv_stamps.argtypes = [C.POINTER(C.c_uint64)]
    L.tg_debug_read_conv_stamps(buf)
    t = np.array(list(buf), dtype=np.float64).reshape(8, 64)
    d_ = np.diff(t[:5], axis=0)
    med = np.median(d_, axis=1)
    if True:                                  # persistent tile-pipelined kernel: consumer wave 0, sums over the workgroup's tiles
        m = np.median(t[:6], axis=1)
        steps = (ci // (64 if PREC == 'bf16' else 32)) * 3
        print("%s: %d tiles per workgroup | wait for the first operands %.0f | K loops %.0f per tile (%.0f per step; at barriers %.0f per step) | epilogue %.0f per tile | "
              "total %.0f per tile (s_memtime ticks)" % (name, m[5], m[0], m[1] / m[5], m[1] / m[5] / steps, m[2] / m[5] / steps, m[3] / m[5], m[4] / m[5]))
