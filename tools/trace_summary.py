#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-iteration picture of one iteration type (tools/mode_trace.py):

    python tools/trace_summary.py <dir with *_kernel_trace.csv> [label]

An iteration = from one k_sample_interp (or k_interp_repack, or the tracking loop's k_track_fwd) dispatch - the first launch of every iteration type - to the next.  Reports, as medians over the last iterations of the trace:
the period (start to start), the sum of kernel durations, the idle time between kernels on the critical stream, and every
kernel's duration / share.  Writes markdown to stdout."""
import collections
import csv
import glob
import os
import statistics
import sys


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip('/'))
    files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        print(f'no kernel trace under {d}')
        return
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Queue_Id', '?')))
    rows.sort()
    # first launch of every iteration: the sampler (not its search-only form <T, 1>, which lk_map_frame runs ahead of the loop)
    starts = [i for i, r in enumerate(rows) if r[2].startswith(('k_interp_repack', 'k_track_fwd')) or (r[2].startswith('k_sample_interp') and not r[2].rstrip().endswith(', 1>'))]
    if len(starts) < 8:
        print('too few iterations in the trace')
        return
    its = []
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        period = rows[b][0] - seg[0][0]
        busy = sum(r[1] - r[0] for r in seg)
        # union of the busy intervals (two streams may overlap)
        cover, cur_s, cur_e = 0, None, None
        for s, e, *_ in sorted(seg):
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    cover += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        cover += cur_e - cur_s
        its.append((period, busy, cover, seg))
    its = its[len(its) // 2:]            # the later half: steady state of the last repeat
    med = lambda xs: statistics.median(xs)
    period, busy, cover = med([i[0] for i in its]), med([i[1] for i in its]), med([i[2] for i in its])
    per = collections.defaultdict(list)
    cnt = collections.defaultdict(list)
    for _, _, _, seg in its:
        c = collections.Counter()
        t = collections.Counter()
        for s, e, n, *_ in seg:
            c[n] += 1
            t[n] += e - s
        for n in t:
            per[n].append(t[n])
            cnt[n].append(c[n])
    print(f'### {label}: {len(its)} iterations')
    print()
    print(f'period {period / 1e3:.1f} us | sum of kernel durations {busy / 1e3:.1f} us | GPU busy (union) {cover / 1e3:.1f} us | idle {max(0, period - cover) / 1e3:.1f} us | dispatches {int(med([len(i[3]) for i in its]))}')
    print()
    print('| kernel | launches / iteration | us / iteration | share of period |')
    print('|---|---|---|---|')
    for n, v in sorted(per.items(), key=lambda kv: -med(kv[1])):
        once = '' if len(v) >= 0.9 * len(its) else f' (only in {len(v)} of {len(its)} iterations: medians over those)'
        print(f'| {n[:60]}{once} | {med(cnt[n]):.0f} | {med(v) / 1e3:.1f} | {100 * med(v) / period:.1f} % |')
    print()
    if len(sys.argv) > 3 and sys.argv[3] == 'gantt':       # one iteration of median length, launch by launch (us from its first launch; queue = stream)
        pick = min(its, key=lambda i: abs(i[0] - period))
        t0 = pick[3][0][0]
        print('| start | end | us | queue | kernel |')
        print('|---|---|---|---|---|')
        for r in pick[3]:
            print(f'| {(r[0] - t0) / 1e3:.1f} | {(r[1] - t0) / 1e3:.1f} | {(r[1] - r[0]) / 1e3:.1f} | {r[3]} | {r[2][:60]} |')
        print()


if __name__ == '__main__':
    main()
