"""Summarise the rocprofv3 --pmc passes of tools/pmc_traffic.sh: mean FETCH_SIZE / WRITE_SIZE (KB) per (kernel, grid size)."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/traffic'
out = collections.defaultdict(lambda: {'FETCH_SIZE': [], 'WRITE_SIZE': []})
for kind in ('fetch', 'write'):
    import os
    paths = sorted(glob.glob('%s/%s/*/*counter_collection.csv' % (root, kind)), key=os.path.getmtime)[-1:]     # the latest pass only
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            name = re.sub(r'\(.*', '', name).replace('void ', '')
            out[(name, int(r['Grid_Size']) // int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for (name, wgs), v in out.items():
    f = sum(v['FETCH_SIZE']) / max(len(v['FETCH_SIZE']), 1)
    w = sum(v['WRITE_SIZE']) / max(len(v['WRITE_SIZE']), 1)
    rows.append((f * len(v['FETCH_SIZE']) + w * len(v['WRITE_SIZE']), name, wgs, len(v['FETCH_SIZE']), f, w))
for tot, name, wgs, n, f, w in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print('%-70s wgs=%6d launches=%3d FETCH_KB=%10.0f WRITE_KB=%10.0f' % (name[:70], wgs, n, f, w))
