#!/usr/bin/env python3
"""Per-kernel medians of the steady iterations inside a rocprofv3 --kernel-trace of tools/slam_run.py (any config): an iteration
is what lies between two consecutive launches of a marker kernel less than 2 ms apart - k_track_final (tracking, the default),
k_bwd_reduce (mapping, stage 'color'), k_adam (mapping, stage 'geometry').

    rocprofv3 --kernel-trace --output-format csv -d DIR -o b -- python tools/slam_run.py --frames 13 --config CFG --out /tmp/x.json
    python tools/trace_track_iter.py DIR [marker]"""
import collections
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + '/**/b_kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_track_final'
marks = [i for i, r in enumerate(rows) if r[2] == marker]
its = []                                       # (period, {kernel: [durations]}) per iteration
for a, b in zip(marks[:-1], marks[1:]):
    p = rows[b][0] - rows[a][0]
    if p > 2_000_000 or b - a > 60:
        continue
    d = collections.defaultdict(list)
    for s, e, n in rows[a + 1:b + 1]:
        d[n].append(e - s)
    its.append((p, d))
its = its[int(len(its) * 0.6):]                # the later frames: steady state (the first mapped frame runs 5x the iterations on a small map)
periods = [p for p, _ in its]
per, cnt = collections.defaultdict(list), collections.defaultdict(list)
for _, d in its:
    for n, v in d.items():
        per[n] += v; cnt[n].append(len(v))
print(f'{marker}: {len(periods)} iterations, median period {statistics.median(periods) / 1e3:.1f} us')
tot = 0.0
for n, v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]) * statistics.median(cnt[kv[0]])):
    m, k = statistics.median(v) / 1e3, statistics.median(cnt[n])
    if len(cnt[n]) < len(periods) // 2:
        continue
    tot += m * k
    print(f'  {n[:60]:60s} x{k:g}  {m:7.1f} us')
print(f'  sum of medians {tot:.1f} us')
