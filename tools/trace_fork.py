#!/usr/bin/env python3
"""Kernel trace of tools/mode_trace.py color -> start / end of every kernel of one steady iteration relative to the iteration's start
(shows what actually overlaps on the two streams)."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Queue_Id', '?')))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('k_interp_repack') or (r[2].startswith('k_sample_interp') and not r[2].rstrip().endswith(', 1>'))]
a, b = starts[-8], starts[-7]
t0 = rows[a][0]
for s, e, n, q in rows[a:b]:
    print(f'{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f})  q{q}  {n[:50]}')
