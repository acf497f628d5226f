"""Diagnostic: workgroup timeline of the generator's transposed-conv forward launch (libtg_stamp.so, built with -DTG_STAMP)."""
import ctypes as C, os, sys, collections
os.environ['TG_LIB'] = 'libtg_stamp.so'
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom
L = lib.load()
L.tg_debug_read_block_times.argtypes = [C.POINTER(C.c_uint64), C.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, hw, ci, co in (("dconv1 8x8x288->128", 8, 288, 128), ("dconv0 4x4x544->256", 4, 544, 256)):
    for tile in (None, "64,64", "64,128"):
        if tile:
            os.environ["TG_IGEMM_TILE"] = tile
        else:
            os.environ.pop("TG_IGEMM_TILE", None)
        x = torch.randn(n, hw, hw, ci, device='cuda'); w = torch.randn(25, co, ci, device='cuda') * 0.05
        y = torch.empty(n, 2 * hw, 2 * hw, co, device='cuda')
        dl = geom.deconv_fwd(n, hw, hw, ci, co)
        dds = lib.desc_array(dl)
        for _ in range(3):
            lib.call("tg_igemm_multi_f32", C.cast(dds, C.c_void_p), len(dds), lib.ptr(x), lib.ptr(w), None, lib.ptr(y), None, 0, lib.cur_stream())
        torch.cuda.synchronize()
        bm, bn = (int(v) for v in tile.split(',')) if tile else (0, 0)
        nb = 8192
        buf = (C.c_uint64 * (3 * nb))()
        L.tg_debug_read_block_times(buf, nb)
        a = np.array(buf[:], dtype=np.uint64).reshape(nb, 3)
        a = a[a[:, 1] > a[:, 0]]
        # keep only the blocks of the last launch: those whose start lies within 1 ms of the latest end
        a = a[a[:, 0] + 100000 > a[:, 1].max()]
        t0 = a[:, 0].min()
        st, en = (a[:, 0] - t0).astype(np.int64), (a[:, 1] - t0).astype(np.int64)
        hw_id = (a[:, 2] & 0xffffffff).astype(np.int64); xcc = (a[:, 2] >> 32).astype(np.int64) & 0xf
        cu = (hw_id >> 8) & 0xf; sh = (hw_id >> 12) & 1; se = (hw_id >> 13) & 0x7
        key = xcc * 1000 + se * 100 + sh * 10 + cu
        dur = (en - st) / 100.0
        per = collections.defaultdict(list)
        for i in range(len(a)):
            per[key[i]].append((st[i] / 100.0, en[i] / 100.0))
        busy = [max(e for _, e in v) for v in per.values()]
        ev = sorted([(s, 1) for s in st] + [(e, -1) for e in en])
        cur = last = area = 0
        for t, dlt in ev:
            area += cur * (t - last); last = t; cur += dlt
        cnt = collections.Counter(len(v) for v in per.values())
        print("%s n=%d tile %-7s blocks %d span %.1f us | block us min %.1f med %.1f max %.1f | CUs used %d, per-CU finish us min %.1f med %.1f max %.1f | mean resident %.0f | blocks/CU %s | late starts (>5us) %d" %
              (name, n, tile or 'model', len(a), en.max() / 100.0, dur.min(), np.median(dur), dur.max(), len(per), min(busy), np.median(busy), max(busy),
               area / float(en.max()), sorted(cnt.items()), int((st > 500).sum())), flush=True)
