"""What the matrix kernels do not cover: reads a rocprofv3 --kernel-trace CSV of `bench.py` and, over a window of whole iterations inside
the timed region, splits the wall time into (a) some MFMA kernel running, (b) only bandwidth / latency kernels running — attributed to
the kernel that has been running longest at that instant — and (c) nothing running (launch gaps).  Usage:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline --soak-seconds 0
    python3 tools/exposed_time.py gpurun_out/tl [first_iteration last_iteration]
"""
import collections, csv, glob, os, re, sys

MFMA = ('conv3x3_pipe_kernel', 'igemm_f32_kernel', 'wgrad3x3_kernel', 'wgrad_f32_kernel', 'packed_fwd', 'packed_wgrad<', 'narrow_dgrad', 'narrow_wgrad<')
MARK = os.environ.get('TG_TRACE_MARK', 'narrow_dgrad')    # a kernel launched PER times per iteration (default: generator image layer, G-update)
PER = int(os.environ.get('TG_TRACE_MARK_PER', '1'))


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:60]


def main():
    d = sys.argv[1]
    path = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    marks = [s for s, e, n in rows if n.startswith(MARK)][::PER]
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 55
    assert len(marks) > hi, (len(marks), hi)
    t0, t1, iters = marks[lo], marks[hi], hi - lo
    ev = []
    for s, e, n in rows:
        if e <= t0 or s >= t1:
            continue
        ev.append((max(s, t0), 1, n, s))
        ev.append((min(e, t1), 0, n, s))
    ev.sort(key=lambda x: (x[0], x[1]))
    running = {}                            # (name, start) -> is_mfma
    last = t0
    mfma_t = idle_t = 0
    exposed = collections.Counter()
    n_mfma_alone = collections.Counter()
    for t, kind, n, s in ev:
        dt = t - last
        if dt > 0:
            if any(running.values()):
                mfma_t += dt
            elif running:
                oldest = min(running, key=lambda k: k[1])
                exposed[oldest[0]] += dt
            else:
                idle_t += dt
        last = t
        if kind == 1:
            running[(n, s)] = n.startswith(MFMA)
        else:
            running.pop((n, s), None)
    tot = t1 - t0
    per = 1e-6 / iters
    print('window: iterations %d..%d, %.3f ms per iteration' % (lo, hi, tot * per))
    print('  an MFMA kernel running      %.3f ms' % (mfma_t * per))
    print('  only other kernels running  %.3f ms' % (sum(exposed.values()) * per))
    print('  nothing running             %.3f ms' % (idle_t * per))
    busy = collections.Counter()
    cnt = collections.Counter()
    for s, e, n in rows:
        if s >= t0 and e <= t1:
            busy[n] += e - s
            cnt[n] += 1
    print('exposed (not under any MFMA kernel) / total kernel time / launches per iteration, by kernel:')
    for n, v in exposed.most_common(30):
        print('  %-62s %.3f / %.3f ms  x%.1f' % (n, v * per, busy[n] * per, cnt[n] / iters))
    print('MFMA kernel time summed over streams: %.3f ms per iteration (running concurrently counts twice)' % (sum(v for n, v in busy.items() if n.startswith(MFMA)) * per))


main()
