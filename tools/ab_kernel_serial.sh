#!/bin/bash
# tools/ab_kernel_serial.sh MODE PATTERN NAME... : per variant ab/lib_NAME.so, the kernels matching PATTERN in a single-stream (LK_SERIAL=1) kernel trace of
# `tools/mode_trace.py MODE 40` - every kernel alone on the chip: average us over the trace
cd "$(dirname "$0")/.."
mode=$1; pat=$2; shift; shift
export TMPDIR=/tmp
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship_ks.so
for v in "$@"; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  rm -rf /tmp/trace_ks
  LK_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_ks -o t -- python tools/mode_trace.py $mode 40 > /tmp/trace_ks.log 2>&1
  python - "$v" "$pat" <<'P'
import csv, glob, sys, collections
v, pat = sys.argv[1], sys.argv[2]
d = collections.defaultdict(list)
for f in glob.glob('/tmp/trace_ks/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if pat in n: d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, t in sorted(d.items()):
    t = sorted(t); print(v, n, 'n', len(t), 'median %.1f us' % t[len(t) // 2], 'mean %.1f' % (sum(t) / len(t)))
P
done
cp /tmp/lib_ship_ks.so loopy_slam_amd/libloopyhip.so
