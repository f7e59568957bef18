#!/bin/bash
# round-6 job J: where the bench step loses what the tracking iteration gained: full bench lines of the two libraries (iterations-only figure, kernels)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship.so
for v in base new base new; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r6j_$v.json
  python - <<P
import json
d = json.load(open('gpurun_out/bench_r6j_$v.json'))
print('$v', 'full %.2f' % d['ms_per_step'], 'iterations %.2f' % d['ms_per_step_iterations'], {k: round(v, 2) for k, v in list(d['kernel_ms_per_step'].items())[:10]})
P
done
cp /tmp/lib_ship.so loopy_slam_amd/libloopyhip.so
