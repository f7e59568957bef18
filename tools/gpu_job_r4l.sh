#!/bin/bash
# mapper composite + loss inside k_decode_bwd: mapper tests, same-box A/B, per-iteration timelines
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_steps_parity.py tests/test_parity_at_size.py tests/test_parity_at_size_configs.py tests/test_slam_api.py tests/test_dist_gloo.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/gpu_tests_map.log
tail -2 gpurun_out/gpu_tests_map.log
ab() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 full %.3f iterations %.3f' % (d['ms_per_step'], d['ms_per_step_iterations']))"; }
for k in 1 2 3 4; do
  LK_MAP_LOSS_INLINE=0 ab composite_launch
  ab composite_in_bwd
done | tee gpurun_out/ab_map_loss.txt
bash tools/gpu_trace_modes.sh r4 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r4.md
grep -E "k_decode_bwd|k_composite|k_loss_rows" gpurun_out/trace_r4.md
