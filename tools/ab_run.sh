#!/bin/bash
# tools/ab_run.sh NAME...  (on the GPU box): bench step time and serial per-kernel ms/step for each ab/lib_NAME.so
cd "$(dirname "$0")/.."
for v in "$@"; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('== $v: %.2f ms/step' % d['ms_per_step'])"
  LK_SERIAL=1 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('   serial %.2f ms/step: ' % d['ms_per_step'] + '  '.join('%s=%.2f' % (n[2:], v) for n, v in list(k.items())[:9]))"
done
