#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> the dispatches around every idle gap of at least MIN_US (default 50) in the last third of the run:
start / end relative to the gap, queue, kernel - what the GPU ran last before it went idle and what it was waiting to be given.

    python tools/trace_boundary.py <dir with *_kernel_trace.csv> [min_gap_us] [n_context]"""
import csv, glob, os, sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]))
rows.sort()
rows = rows[2 * len(rows) // 3:]
end = rows[0][1]
shown = 0
for k in range(1, len(rows)):
    s, e, q, n = rows[k]
    if s > end and (s - end) / 1e3 >= min_gap and shown < 12:
        shown += 1
        print(f'--- gap {(s - end) / 1e3:.1f} us')
        for j in range(max(0, k - ctx), min(len(rows), k + ctx)):
            ss, ee, qq, nn = rows[j]
            print(f'   {"*" if j == k else " "} start {(ss - s) / 1e3:9.1f}  end {(ee - s) / 1e3:9.1f}  q{qq:>3s}  {nn}')
    end = max(end, e)
