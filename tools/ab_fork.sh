#!/bin/bash
# tools/ab_fork.sh NAME...  (on the GPU box): per variant ab/lib_NAME.so, three overlapped bench steps and the colour-iteration timeline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  for k in 1 2 3; do
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('== $v: %.2f ms/step' % d['ms_per_step'])"
  done
  rm -rf /tmp/trc; rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -o t -- python tools/mode_trace.py color 40 > /dev/null 2>&1
  python tools/trace_fork.py /tmp/trc
done
