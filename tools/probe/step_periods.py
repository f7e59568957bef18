#!/usr/bin/env python3
"""Per-iteration periods of the mapping call inside a whole benchmark step (rocprofv3 kernel trace of tools/probe/step_trace.py): start of every
iteration's first launch to the start of the next one's, for the last complete step of the trace.

    python tools/probe/step_periods.py <trace dir>"""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
first = [i for i, r in enumerate(rows) if r[3].startswith('k_interp_repack') or r[3].startswith('k_sample_interp<8, 2>')]
# split into calls: a gap of more than 2 ms between iteration starts
calls, cur = [], [first[0]]
for a, b in zip(first[:-1], first[1:]):
    if rows[b][0] - rows[a][0] > 2_000_000:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
call = [c for c in calls if len(c) == 60][-2]
per = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(call[:-1], call[1:])]
kind = ['geo' if rows[i][3].startswith('k_sample_interp') else 'color' for i in call]
print('mapping call of', len(call), 'iterations; periods in us, in order:')
print('geo  :', ' '.join(f'{p:.0f}' for p, k in zip(per, kind) if k == 'geo'))
print('color:', ' '.join(f'{p:.0f}' for p, k in zip(per, kind) if k == 'color'))
# the call's span: from the k_row_rank in front of it to the last launch before the next tracking call's k_pregather
i0 = max(i for i in range(call[0]) if rows[i][3].startswith('k_row_rank'))
end = max(r[1] for r in rows[call[-1]:call[-1] + 12] if not r[3].startswith('k_pregather'))
print(f'k_row_rank -> first iteration: {(rows[call[0]][0] - rows[i0][0]) / 1e3:.0f} us; first iteration -> end of the call: {(end - rows[call[0]][0]) / 1e3:.0f} us; '
      f'sum of the {len(per)} periods {sum(per):.0f} us')
