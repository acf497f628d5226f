"""(Round 1-2 tool: TG_IGEMM_TILE is read once at library load since round 3, so the in-process sweep below no longer switches tiles —
use tools/tile_audit_step.sh, which runs one process per tile.)
Audit of the igemm tile model: every single-problem conv-shaped launch of the bench line (profiles/r01_launches.csv) timed with each
tile candidate (TG_IGEMM_TILE) against the model's own pick; prints what a perfect per-shape choice would save per step."""
import collections, csv, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from tg import lib, geom

lib.load()
TILES = ["128,128", "64,128", "64,64", "128,64", "32,128", "128,32"]
shapes = collections.Counter()
for r in csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r01_launches.csv'))):
    m = re.match(r"M=1x(\d+) N=(\d+) K=(\d+)x(\d+) in=(\d+)x(\d+) s=(\d+) os=1$", r['desc'] or '')
    if m and r['class'] == 'igemm_f32':
        shapes[tuple(int(v) for v in m.groups())] += 1


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


tot_model = tot_best = 0.0
rows = []
for (M, N, taps, ld, h, w, s), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3]):
    k = {1: 1, 9: 3, 25: 5}.get(taps)
    if k is None:
        continue
    d = None
    for pad in ('SAME', 'VALID'):
        g0 = geom.conv_fwd(1, h, w, 32, 32, k, s, pad)
        ho, wo = g0.h_out, g0.w_out
        if ho > 0 and wo > 0 and M % (ho * wo) == 0:
            n = M // (ho * wo)
            d = geom.conv_fwd(n, h, w, ld, N, k, s, pad)
            break
    if d is None:
        continue
    x = torch.randn(n, h, w, ld, device='cuda')
    wt = torch.randn(N, taps, ld, device='cuda') * 0.05
    y = torch.empty(n, d.h_out, d.w_out, N, device='cuda')
    st = lib.cur_stream()
    call = lambda: lib.call("tg_igemm_f32", d, lib.ptr(x), lib.ptr(wt), None, lib.ptr(y), None, 0, st)
    os.environ.pop("TG_IGEMM_TILE", None)
    t_model = timeit(call)
    res = {}
    for t in TILES:
        os.environ["TG_IGEMM_TILE"] = t
        try:
            res[t] = timeit(call)
        except lib.TgError:
            pass
    os.environ.pop("TG_IGEMM_TILE", None)
    t_model = min(t_model, timeit(call))                 # again after the sweep: the first timing of a shape includes clock ramp-up
    best = min(res, key=res.get)
    tot_model += cnt * t_model
    tot_best += cnt * min(res[best], t_model)
    rows.append((cnt * (t_model - min(res[best], t_model)), (M, N, taps, ld, h, s), cnt, t_model, best, res[best]))
for save, shp, cnt, tm, best, tb in sorted(rows, reverse=True)[:18]:
    print("M=%d N=%d K=%dx%d in=%d s=%d  x%d  model %.4f ms  best %-8s %.4f ms  saves %.4f ms/step" % (shp + (cnt, tm, best, tb, save)), flush=True)
print("total over %d shapes: model %.3f ms/step, best-per-shape %.3f ms/step" % (len(rows), tot_model, tot_best))
