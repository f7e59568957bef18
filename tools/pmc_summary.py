#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (sum over dispatches / dispatch count)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k].add(r['Dispatch_Id'])
names = sorted({c for k in agg for c in agg[k]})
print('kernel'.ljust(28), 'n'.rjust(5), ' '.join(n[-18:].rjust(18) for n in names))
key = 'GRBM_GUI_ACTIVE' if 'GRBM_GUI_ACTIVE' in names else 'SQ_WAVE_CYCLES' if 'SQ_WAVE_CYCLES' in names else ('SQ_INSTS_VMEM_RD' if 'SQ_INSTS_VMEM_RD' in names else names[0])
for k in sorted(agg, key=lambda k: -agg[k].get(key, 0)):
    if not k.startswith(('k_', 'void k_')):
        continue
    n = len(cnt[k])
    print(k[:28].ljust(28), str(n).rjust(5), ' '.join(f'{agg[k].get(c, 0) / n:18.0f}' for c in names))
