#!/usr/bin/env python3
"""The launches around the geometry -> colour stage switch of a mapping call inside a benchmark step (trace of tools/probe/step_trace.py)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]))
rows.sort()
# the last 'geometry' iteration of a call: a k_sample_interp<8, 2> whose next iteration-start is a k_interp_repack
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_interp_repack") or r[3].startswith("k_sample_interp<8, 2>")]
idx = [a for a, b in zip(starts[:-1], starts[1:]) if rows[a][3].startswith("k_sample_interp") and rows[b][3].startswith("k_interp_repack")]
k = idx[len(idx) // 2]
t0 = rows[k][0]
end = t0
for s, e, q, n in rows[k:k + 30]:
    gap = (s - end) / 1e3
    tag = ("gap %6.1f" % gap) if gap > 0 else ("ovl %6.1f" % -gap)
    print("%8.1f %8.1f dur %7.1f %s q%2s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, tag, q, n))
    end = max(end, e)
