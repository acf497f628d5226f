"""Diagnostic: per-iteration launch counts of kernels matching a pattern in a rocprofv3 --kernel-trace CSV (iterations delimited by every third
`step_inc` launch).  python3 tools/debug_trace_counts.py <trace dir> <pattern>"""
import csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path)))
marks = [t for t, n in rows if 'step_inc' in n][::3]
counts = [0] * (len(marks) + 1)
k = 0
for t, n in rows:
    while k < len(marks) and t >= marks[k]:
        k += 1
    if sys.argv[2] in n:
        counts[k] += 1
print('iterations', len(marks), 'launches of', sys.argv[2], 'per iteration:', counts)
