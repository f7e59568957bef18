#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> the dispatches of a window of the run, in start order with queue ids, gaps and overlaps:
    python tools/trace_window.py <dir> <anchor kernel substring> [occurrence (default: the middle one)] [n dispatches after it (default 30)]"""
import csv, glob, os, sys
d, anchor = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]))
rows.sort()
hits = [i for i, r in enumerate(rows) if anchor in r[3]]
k = hits[len(hits) // 2] if occ < 0 else hits[occ]
t0 = rows[k][0]
end = rows[k][0]
for s, e, q, name in rows[k:k + n]:
    gap = (s - end) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}  {"gap %6.1f" % gap if gap > 0 else "ovl %6.1f" % -gap}  q{q:>3s}  {name}')
    end = max(end, e)
