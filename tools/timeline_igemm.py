"""Diagnostic: workgroup timeline of one igemm launch (start/end in 10 ns ticks, CU ids) from libtg_stamp.so."""
import ctypes as C, os, sys, collections
os.environ['TG_LIB'] = 'libtg_stamp.so'
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom
L = lib.load()
N, hw, ci, co = 250, 32, 128, 128
x = torch.randn(N, hw, hw, ci, device='cuda'); w = torch.randn(co, 9, ci, device='cuda') * 0.05
d = geom.conv_fwd(N, hw, hw, ci, co, 3, 1, 'SAME'); y = torch.empty(N, hw, hw, co, device='cuda')
for _ in range(3):
    lib.call("tg_igemm_f32", d, lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, lib.cur_stream())
torch.cuda.synchronize()
nb = 2000
buf = (C.c_uint64 * (3 * nb))()
L.tg_debug_read_block_times.argtypes = [C.POINTER(C.c_uint64), C.c_int]
L.tg_debug_read_block_times(buf, nb)
a = np.array(buf[:], dtype=np.uint64).reshape(nb, 3)
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0).astype(np.int64), (a[:, 1] - t0).astype(np.int64)
hw_id = (a[:, 2] & 0xffffffff).astype(np.int64); xcc = (a[:, 2] >> 32).astype(np.int64) & 0xf
cu = (hw_id >> 8) & 0xf; sh = (hw_id >> 12) & 1; se = (hw_id >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 10 + cu
print("kernel span %.1f us; block duration us: min %.1f median %.1f max %.1f" % (en.max() / 100.0, (en - st).min() / 100.0, np.median(en - st) / 100.0, (en - st).max() / 100.0))
print("distinct CU keys", len(set(key.tolist())), "xcc values", sorted(set(xcc.tolist())))
per = collections.defaultdict(list)
for i in range(nb):
    per[key[i]].append((st[i], en[i], i))
for k in list(per)[:3]:
    print("CU", k, [(round(s / 100.0, 1), round(e / 100.0, 1), i) for s, e, i in sorted(per[k])])
cnt = collections.Counter(len(v) for v in per.values())
print("blocks per CU histogram", sorted(cnt.items()))
# concurrency over time
ev = sorted([(s, 1) for s in st] + [(e, -1) for e in en])
cur = 0; last = 0; area = 0
for t, dlt in ev:
    area += cur * (t - last); last = t; cur += dlt
print("mean resident workgroups %.1f" % (area / float(en.max())))
