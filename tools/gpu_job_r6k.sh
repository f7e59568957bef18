#!/bin/bash
# round-6 job K: the shipped library (HEAD, fused tracker forward dropped) against the library job J called "base": full bench lines alternating,
# then the per-iteration timelines of the three iteration types on the shipped library
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship.so
for v in base head base head; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 > gpurun_out/bench_r6k_$v.json
  python - <<P
import json
d = json.load(open('gpurun_out/bench_r6k_$v.json'))
print('$v', 'full %.2f' % d['ms_per_step'], 'iterations %.2f' % d['ms_per_step_iterations'], {k: round(v, 2) for k, v in list(d['kernel_ms_per_step'].items())[:10]})
P
done
cp /tmp/lib_ship.so loopy_slam_amd/libloopyhip.so
bash tools/gpu_trace_modes.sh r6k > /dev/null
for m in track geo color; do python tools/trace_summary.py /tmp/trace_$m "$m" gantt > gpurun_out/gantt_r6k_$m.md 2>/dev/null; done
grep -E "period|^## " gpurun_out/trace_r6k.md | head -20
