#!/bin/bash
# tools/ab_full.sh N NAME... : N alternating bench runs per variant (ab/lib_NAME.so): full step, iterations alone, largest kernels
cd "$(dirname "$0")/.."
n=$1; shift
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship_ab.so
for k in $(seq $n); do for v in "$@"; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', 'full %.2f' % d['ms_per_step'], 'iterations %.2f' % d['ms_per_step_iterations'], {k: round(v, 2) for k, v in list(d['kernel_ms_per_step'].items())[:8]})"
done; done
cp /tmp/lib_ship_ab.so loopy_slam_amd/libloopyhip.so
