"""Summarise tools/pmc_conv3x3_bf16.sh (rocprofv3 --pmc passes a, b + a kernel trace t on tools/bench_conv3x3_bf16.py): per launch shape of the
halo-tiled 3x3 kernel — duration, clock, matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over 4 SIMDs x 256 CUs x cycles), wait /
active shares of the wave cycles, LDS bank conflicts.  GRBM_GUI_ACTIVE is summed over the 8 XCDs (-> / 8)."""
import collections, csv, glob, os, re, sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_c3_bf16'


def latest(sub, pat):
    return sorted(glob.glob('%s/%s/*/*%s' % (root, sub, pat)), key=os.path.getmtime)[-1]


def key_of(r):
    m = re.search(r'((?:conv3x3_\w+|wgrad3x3)_kernel<[^>]*>)', r['Kernel_Name'])
    if not m:
        return None
    gx = int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size'])
    wx = int(r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else r['Workgroup_Size'])
    return (m.group(1), gx // wx)


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(latest(sub, 'counter_collection.csv'))):
        k = key_of(r)
        if k:
            out[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


dur = collections.defaultdict(list)
for r in csv.DictReader(open(latest('t', 'kernel_trace.csv'))):
    k = key_of(r)
    if k:
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
a, b = counters('a'), counters('b')
print("%-46s %5s %7s %7s %9s %8s %13s %8s %12s" % ('kernel', 'wgs', 'dur_us', 'clk_GHz', 'MFMA_busy', 'WAIT_ANY', 'WAIT_INST_ANY', 'ACTIVE', 'LDS_conflict'))
for key in sorted(a, key=lambda k: -a[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
    ca, cb = a[key], b.get(key, {})
    ds = sorted(dur.get(key, [0]))
    d = ds[len(ds) // 2]
    gui = cb.get('GRBM_GUI_ACTIVE', 0) / 8.0
    clk = gui / (d * 1e3) if d else 0
    busy = ca.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * 256 * gui) if gui else 0
    wc = ca.get('SQ_WAVE_CYCLES', 1) or 1
    print("%-46s %5d %7.1f %7.2f %9.3f %8.2f %13.2f %8.2f %12.3f" % (key[0], key[1], d, clk, busy, ca.get('SQ_WAIT_ANY', 0) / wc, ca.get('SQ_WAIT_INST_ANY', 0) / wc,
          ca.get('SQ_ACTIVE_INST_ANY', 0) / wc, ca.get('SQ_LDS_BANK_CONFLICT', 0) / max(ca.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
