import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + '/**/b_kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
rows.sort()
# frames: delimited by k_pregather<8> launches of the tracker (one per tracked frame)
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_pregather<8>')]
print(len(rows), 'kernels', len(marks), 'pregather<8> marks')
per = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    if any(s[2].startswith('k_bwd_reduce') or s[2].startswith('k_adam') for s in seg):
        continue        # a mapped frame
    period = rows[b][0] - rows[a][0]
    busy = 0; last = seg[0][0]
    gaps = []
    for s, e, n in seg:
        if s > last:
            gaps.append((s - last, n))
        busy += max(0, e - max(s, last)); last = max(last, e)
    per.append((period, busy, gaps, [n for _, _, n in seg]))
per = per[len(per) // 2:]
print('tracked frames:', len(per), 'median period %.1f us busy %.1f us' % (statistics.median(p[0] for p in per) / 1e3, statistics.median(p[1] for p in per) / 1e3))
p = per[len(per) // 2]
big = sorted(p[2], reverse=True)[:12]
print('largest gaps of one frame (us, kernel that follows):', [(round(g / 1e3, 1), n[:40]) for g, n in big])
cnt = collections.Counter(n for n in p[3] if not n.startswith('k_'))
print('non-library kernels in the frame:', dict(cnt))
first = [n for n in p[3][:30]]
print('first kernels:', first)
