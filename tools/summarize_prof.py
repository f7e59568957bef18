#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a bench run from gpurun_out/ into profiles/ (tracked).

    python tools/summarize_prof.py r1          # reads gpurun_out/prof_r1_{stats,fetch,write}/
"""
import collections
import csv
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
src_tag = sys.argv[2] if len(sys.argv) > 2 else tag          # gpurun_out/prof_<src_tag>_*
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out')
dst = os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, f'prof_{src_tag}_stats', 'b_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
lines = [f'# rocprofv3 summary, round tag {tag}', '',
         'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only`',
         '(kernel table = profiles/%s_kernel_stats.csv; durations in ns).' % tag, '',
         '| kernel | calls | avg us | % |', '|---|---|---|---|']
for r in list(csv.DictReader(open(os.path.join(dst, f'{tag}_kernel_stats.csv'))))[:24]:
    lines.append(f"| {r['Name'].split('(')[0].replace('void ', '')} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
lines += ['', '## HBM traffic (PMC, separate passes: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with `--kernel-trace` only)', '',
          'Counter unit = KiB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950 reports half of the bytes of',
          'wide coalesced reads: the "read (corrected)" column doubles it; WRITE_SIZE is uncalibrated there and quoted as is.', '',
          '| kernel | launches | FETCH_SIZE avg KiB | read (corrected) MB | WRITE_SIZE avg KiB |', '|---|---|---|---|---|']
agg = {}
for cname, d in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(src, f'prof_{src_tag}_{d}', 'b_counter_collection.csv'))):
        if r['Counter_Name'] == cname:
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            a[k][0] += 1
            a[k][1] += float(r['Counter_Value'])
    agg[cname] = a
for k, (n, v) in sorted(agg['FETCH_SIZE'].items(), key=lambda kv: -kv[1][1])[:24]:
    if not k.startswith('k_'):
        continue
    w = agg['WRITE_SIZE'].get(k, [1, 0.0])
    lines.append(f'| {k} | {n} | {v / n:.0f} | {2 * v / n * 1024 / 1e6:.1f} | {w[1] / max(w[0], 1):.0f} |')
open(os.path.join(dst, f'{tag}_rocprof_summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
