import csv,sys,glob
f=glob.glob('/tmp/prof/**/b_kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:26]:
    print(f"{r['Name'].split('(')[0][:34]:34s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}  tot_ms {float(r['TotalDurationNs'])/1e6:7.2f}")
