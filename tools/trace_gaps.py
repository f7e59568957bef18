#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> the largest idle gaps of the GPU (no kernel of any stream running) inside the traced run, with the
kernels on either side: where a loop stalls (waiting for another stream, for the host, for a read-back).

    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [min_gap_us] [skip_first_n_kernels]"""
import csv, glob, os, sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
rows = rows[len(rows) // 2:]                      # second half: steady repeats
end, last = rows[0][1], rows[0][2]
gaps, busy = [], 0
t0 = rows[0][0]
for s, e, n in rows[1:]:
    if s > end:
        g = (s - end) / 1e3
        if g >= min_gap:
            gaps.append((g, last, n, (s - t0) / 1e3))
    if e > end:
        end, last = e, n
span = (rows[-1][1] - t0) / 1e3
print(f'span {span / 1e3:.2f} ms, {len(rows)} kernels, idle gaps >= {min_gap} us: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e3:.2f} ms')
for g, a, b, t in sorted(gaps, reverse=True)[:25]:
    print(f'  {g:8.1f} us at {t / 1e3:8.2f} ms   after {a[:40]:40s} before {b[:40]}')
