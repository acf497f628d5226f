#!/usr/bin/env python3
"""After tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, separate, no trace domains): write the HBM-traffic
record bench.py's `roofline.traffic` is read from.  The record carries the sha256 of the kernel sources it was collected for (csrc/igemm.hip +
csrc/conv3x3_bf16.hip + csrc/wgrad3x3.hip, concatenated); bench.py reports traffic = null for any other source (a stale figure is worse than none).

    python tools/make_traffic_json.py [gpurun_out/traffic] [round tag] > profiles/<round>_traffic.json
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/traffic'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r03'
vals = collections.defaultdict(lambda: {'FETCH_SIZE': [], 'WRITE_SIZE': []})
for kind in ('fetch', 'write'):
    paths = sorted(glob.glob('%s/%s/*/*counter_collection.csv' % (root, kind)), key=os.path.getmtime)[-1:]
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            name = re.sub(r'\(.*', '', name).replace('void ', '')
            vals[(name, int(r['Grid_Size']) // int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))


def mean(v):
    return sum(v) / max(len(v), 1)


def entry(match, wgs, min_write_kb=0):
    """mean counters of the launches of one (kernel, grid).  A persistent kernel has the same grid for every problem size: `min_write_kb`
    keeps the launches of one size (the two passes run the same program, so launch i of the FETCH_SIZE pass is launch i of the
    WRITE_SIZE pass)."""
    for (name, w), v in vals.items():
        if (name.endswith(match) or name.endswith(match[:-1] + ', false>')) and w == wgs:      # trailing defaulted template argument (STAT2 = false)
            fs, ws = v['FETCH_SIZE'], v['WRITE_SIZE']
            keep = [i for i in range(min(len(fs), len(ws))) if ws[i] >= min_write_kb]
            if not keep:
                return None
            f, wr = mean([fs[i] for i in keep]), mean([ws[i] for i in keep])
            return dict(kernel_name=name, workgroups=w, launches_profiled=len(keep), FETCH_SIZE_KB_raw=round(f), WRITE_SIZE_KB=round(wr),
                        traffic_bytes_corrected=int(round((2 * f + wr) * 1024)), traffic_bytes_uncorrected=int(round((f + wr) * 1024)))
    return None


M, C = 250 * 32 * 32, 128
alg_fwd = 4 * (M * C + M * C + C * 9 * C)                       # input + output + filter, fp32
dom = entry('conv3x3_pipe_kernel<32, true, false>', 256, min_write_kb=100000)      # the 250-image launches (131 MB written); the 128-image heads of split launches share the grid
CSRC = os.path.join(ROOT, 'tensorflow-implementation-of-triple-gan_amd', 'csrc')
SOURCES = ('igemm.hip', 'conv3x3_bf16.hip', 'wgrad3x3.hip')        # every kernel bench.py's roofline object names (bench.TRAFFIC_SOURCES)
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) on `python3 bench.py --steps 3 --warmup 2 --no-graph`, "
              "round %s, one MI355X; tools/pmc_traffic.sh + tools/make_traffic_json.py" % tag,
    "units": "counter values are KB; gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams -> x2; WRITE_SIZE exact",
    "kernel_source_files": list(SOURCES),
    "kernel_sources_sha256": hashlib.sha256(b''.join(open(os.path.join(CSRC, f), 'rb').read() for f in SOURCES)).hexdigest(),
}
if dom:
    n_act = 12.0 / 30.0                                          # share of tg_igemm_actsum launches (they also read the producing layer's activation)
    dom.update(kernel="conv3x3_pipe_kernel<W=32,COLSUM,fp32>, 256 persistent workgroups walking 1000 tiles, 250 images 32x32x128 -> 128: classifier conv1_2 / conv1_3 "
                      "forward with fused mean-only-BN column sums and their input gradients with the fused activation derivative + column sums (tg_igemm_actsum)",
               algorithmic_bytes=int(alg_fwd + n_act * 4 * M * C))
    out["dominant_launch"] = dom
head = entry('conv3x3_pipe_kernel<32, true, false>', 256)
if head and dom and head['launches_profiled'] > dom['launches_profiled']:
    out["all_launches_of_that_kernel"] = dict(head, note="250-image launches and the 128-image heads of the D-update's split 130-image launches together")
wg = entry('wgrad3x3_kernel<32, false>', 256)
if wg:
    wg.update(kernel="wgrad3x3_kernel<W=32,fp32>, 256 workgroups (4 channel chunks x 64 pixel splits): filter gradient of conv1_2 / conv1_3, activation tile read once for the nine taps",
              algorithmic_bytes=int(4 * (2 * M * C + 64 * 9 * C * C)))
    out["wgrad_launch"] = wg
for nm, key, wgs in (("mobn_apply", "mobn_apply", None), ("mobn_center", "mobn_center", None)):
    best = None
    for (name, w), v in vals.items():
        if key in name and (best is None or mean(v['WRITE_SIZE']) > best[1]):
            best = (w, mean(v['WRITE_SIZE']), mean(v['FETCH_SIZE']))
    if best:
        out.setdefault("bandwidth_bound_passes", {})[nm] = dict(workgroups=best[0], FETCH_SIZE_KB_raw=round(best[2]), WRITE_SIZE_KB=round(best[1]))
json.dump(out, sys.stdout, indent=1)
print()
