#!/usr/bin/env python3
"""One iteration of a rocprofv3 --kernel-trace of bench.py as a two-queue timeline: start (us), duration (us), queue, kernel — with the three
solver runs delimited by their optimiser step markers (step_inc: D, G, C).  Usage: tools/timeline_iteration.py <kernel_trace.csv> [iteration]."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
inc = [i for i, r in enumerate(rows) if 'step_inc' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(inc) // 3 // 2


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:64]


marks = [inc[3 * k - 1], inc[3 * k], inc[3 * k + 1], inc[3 * k + 2]]
seg = rows[marks[0]:marks[3] + 1]
t0 = int(seg[0]['Start_Timestamp'])
queues = sorted(set(r['Queue_Id'] for r in seg))
print("# iteration %d of %s: %.3f ms" % (k, sys.argv[1].split('/')[-1], (int(seg[-1]['End_Timestamp']) - t0) / 1e6))
for a, b, name in zip(marks[:-1], marks[1:], 'DGC'):
    print("# %s-update: %.3f ms, %d launches" % (name, (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e6, b - a))
print("# start_us  dur_us queue kernel")
for r in seg:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print("%9.1f %7.1f q%d %s" % (s, e - s, queues.index(r['Queue_Id']), short(r['Kernel_Name'])))
