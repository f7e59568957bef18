#!/bin/bash
# tools/ab_quick.sh N NAME...  : N alternating bench runs per variant (overlapped step time only)
cd "$(dirname "$0")/.."
n=$1; shift
for k in $(seq $n); do for v in "$@"; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v %.2f' % d['ms_per_step'])"
done; done
