"""Reads the per-launch tables written by tools/tile_audit_step.sh: for every generic-kernel launch shape of the step, the model's tile and
time beside each forced tile; the sum of what a perfect per-shape choice would save per iteration."""
import collections, csv, glob, os, re, sys

O = sys.argv[1]


def table(path):
    t = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['class'] != 'igemm_f32' or 'tile=0x0' in r['desc'] or not r['desc'].startswith('M='):
            continue
        shape = r['desc'].split(' tile=')[0]
        tile = re.search(r'tile=(\d+x\d+)', r['desc']).group(1)
        e = t.setdefault(shape, [0, 0.0, tile, float(r['gflop'])])
        e[0] += 1
        e[1] += float(r['ms'])
    return t


model = table(os.path.join(O, 'model.csv'))
forced = {os.path.basename(p)[5:-4]: table(p) for p in sorted(glob.glob(os.path.join(O, 'tile_*.csv')))}
save = 0.0
rows = []
for shape, (n, ms, tile, gf) in model.items():
    alts = {}
    for name, t in forced.items():
        if shape in t and t[shape][2] == name.replace(',', 'x'):
            alts[name] = t[shape][1] / t[shape][0] * n
    best = min(alts, key=alts.get) if alts else None
    gain = ms - alts[best] if best and alts[best] < ms else 0.0
    save += gain
    rows.append((gain, shape, n, tile, ms, ' '.join('%s=%.3f' % (k, v / 2.0) for k, v in sorted(alts.items(), key=lambda kv: kv[1]))))
iters = 2.0
for gain, shape, n, tile, ms, alts in sorted(rows, reverse=True)[:40]:
    print('%-52s x%d model %-7s %.3f ms | %s | saves %.3f' % (shape, n / iters, tile, ms / iters, alts, gain / iters))
print('a perfect per-shape tile would save %.3f ms per iteration' % (save / iters))
