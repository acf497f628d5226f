"""Summarise tools/pmc_igemm.sh (rocprofv3 --pmc passes a, b + a kernel trace t on tools/bench_igemm.py): per forward launch shape of
igemm_f32_kernel — duration, clock, matrix-pipe busy fraction, wait / active shares, LDS bank conflicts."""
import collections, csv, glob, os, re, sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'


def latest(sub, pat):
    return sorted(glob.glob('%s/%s/*/*%s' % (root, sub, pat)), key=os.path.getmtime)[-1]


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(latest(sub, 'counter_collection.csv'))):
        if 'igemm_f32_kernel' not in r['Kernel_Name']:
            continue
        tile = re.search(r'igemm_f32_kernel<(\d+), (\d+)', r['Kernel_Name']).groups()
        out[(tile, int(r['Grid_Size']) // int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


dur = collections.defaultdict(list)
for r in csv.DictReader(open(latest('t', 'kernel_trace.csv'))):
    if 'igemm_f32_kernel' in r['Kernel_Name']:
        tile = re.search(r'igemm_f32_kernel<(\d+), (\d+)', r['Kernel_Name']).groups()
        wgs = int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size']) // int(r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else r['Workgroup_Size'])
        dur[(tile, wgs)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
a, b = counters('a'), counters('b')
print("tile      wgs   dur_us  clk_GHz  MFMA_BUSY/(4 SIMD x 256 CU x cycles)  WAIT_ANY/WAVE  WAIT_INST_ANY/WAVE  ACTIVE/WAVE  LDS_CONFLICT/LDS_ACTIVE  MFMA_MOPS_F32")
for key in sorted(a, key=lambda k: -a[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
    ca, cb = a[key], b.get(key, {})
    d = sorted(dur.get(key, [0]))[len(dur.get(key, [0])) // 2]
    gui = cb.get('GRBM_GUI_ACTIVE', 0) / 8.0             # the counter is summed over the 8 XCDs
    clk = gui / (d * 1e3) if d else 0
    busy = ca.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * 256 * gui) if gui else 0
    wc = ca.get('SQ_WAVE_CYCLES', 1) or 1
    print("%-9s %5d %7.1f  %5.2f    %.3f %36s %.2f %13s %.2f %17s %.2f %10s %.3f %20s %.3g" % (
        'x'.join(key[0]), key[1], d, clk, busy, '', ca.get('SQ_WAIT_ANY', 0) / wc, '', ca.get('SQ_WAIT_INST_ANY', 0) / wc, '',
        ca.get('SQ_ACTIVE_INST_ANY', 0) / wc, '', ca.get('SQ_LDS_BANK_CONFLICT', 0) / max(ca.get('SQ_LDS_IDX_ACTIVE', 1), 1), '',
        cb.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0)))
