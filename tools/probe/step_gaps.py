#!/usr/bin/env python3
"""Where a benchmark step's time goes besides its iterations: from a rocprofv3 kernel trace of tools/probe/step_trace.py, the last full steps
(a step = from one k_pregather of a TRACKING call to the next), the union of busy time over all queues, and every idle gap of the device
above 8 us with the launches on either side.

    python tools/probe/step_gaps.py <trace dir>"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]))
rows.sort()
# a tracking call starts with k_pregather<8> followed (soon) by k_track_final; a mapping call's k_pregather is followed by searches
pg = [i for i, r in enumerate(rows) if r[3].startswith('k_pregather')]
starts = [i for i in pg if any(rows[j][3].startswith('k_track_final') for j in range(i, min(i + 4, len(rows))))]
if len(starts) < 4:
    print('too few steps in the trace'); sys.exit(0)
tot = []
for a, b in zip(starts[-4:-1], starts[-3:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    cover, end = 0, t0
    gaps = []
    prev = None
    for s, e, q, n in seg:
        if s > end:
            if s - end > 8000 and prev is not None:
                gaps.append(((s - end) / 1e3, prev, n, (end - t0) / 1e3))
            cover += 0
            end_new = e
        if e > end:
            cover += e - max(s, end)
            if e > end:
                prev = n
            end = e
    tot.append(((t1 - t0) / 1e3, cover / 1e3, gaps))
for k, (per, busy, gaps) in enumerate(tot):
    print(f'step {k}: period {per:.0f} us, device busy (union of queues) {busy:.0f} us, idle {per - busy:.0f} us; idle gaps > 8 us: {sum(g[0] for g in gaps):.0f} us in {len(gaps)}')
per, busy, gaps = tot[-1]
print('gaps of the last step (us into the step, length, after -> before):')
for g in gaps:
    print(f'  {g[3]:9.0f}  {g[0]:7.1f}  {g[1]} -> {g[2]}')
