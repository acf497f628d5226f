"""Diagnostic: the two sums of tg_igemm_bnbwdstat_f32 against float64 sums of its own output, per (segment, channel)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom
lib.load()
st = lib.cur_stream
for (n, hw, ci, co, segs) in [(5, 8, 256, 512, [3, 2]), (7, 6, 64, 96, [3, 4]), (9, 16, 128, 256, [9]), (6, 16, 128, 128, [2, 4])]:
    rng = np.random.default_rng(22)
    g_in = torch.from_numpy(rng.standard_normal((n, hw, hw, ci)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((co, 9, ci)) * 0.05).astype(np.float32)).cuda()
    xbn = torch.from_numpy(rng.standard_normal((n, hw, hw, co)).astype(np.float32)).cuda()
    d = geom.conv_fwd(n, hw, hw, ci, co, 3, 1, 'SAME')
    seg_rows = [s * hw * hw for s in segs]
    sa = (C.c_int32 * len(segs))(*seg_rows)
    dy = torch.zeros((n, hw, hw, co), device='cuda')
    sums = torch.zeros(8 * len(segs) * 2 * co, dtype=torch.float64, device='cuda')
    lib.call_igemm('tg_igemm_bnbwdstat_f32', d, lib.ptr(g_in), lib.ptr(w), lib.ptr(xbn), lib.ptr(dy), sa, len(segs), lib.ptr(sums), 0, st())
    torch.cuda.synchronize()
    got = sums.reshape(8, len(segs), 2, co).sum(0).cpu().numpy()
    rows = n * hw * hw
    dyd, xd = dy.double().reshape(rows, co).cpu().numpy(), xbn.double().reshape(rows, co).cpu().numpy()
    r0 = 0
    for i, r in enumerate(seg_rows):
        s0, s1 = dyd[r0:r0 + r].sum(0), (dyd[r0:r0 + r] * xd[r0:r0 + r]).sum(0)
        print((n, hw, ci, co), 'seg', i, 'S0 err', np.abs(got[i, 0] - s0).max(), 'S1 err', np.abs(got[i, 1] - s1).max(), 'scale', np.abs(s1).max(),
              'bad cols', np.nonzero(np.abs(got[i, 1] - s1) > 1e-2)[0][:12])
        r0 += r
