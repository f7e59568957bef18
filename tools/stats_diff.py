import csv, sys, glob
def load(d):
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
    return {r['Name'].split('(')[0].replace('void ', '')[:44]: (int(r['Calls']), float(r['TotalDurationNs']) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3])
print('%-44s %8s %8s %9s %9s' % ('kernel', 'calls/st', 'calls/st', 'ms/step', 'ms/step'))
tot = [0, 0]
for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1])):
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    tot[0] += ta; tot[1] += tb
    if abs(tb - ta) / steps > 0.02 or ca != cb:
        print('%-44s %8.1f %8.1f %9.3f %9.3f' % (k, ca / steps, cb / steps, ta / steps, tb / steps))
print('sum of kernel time per step', tot[0] / steps, tot[1] / steps)
