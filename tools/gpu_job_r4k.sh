#!/bin/bash
# tracker loss inside the decoder launches: tracker tests, A/B of the two switches on one box, tracking-iteration timeline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_steps_parity.py tests/test_parity_at_size.py tests/test_parity_at_size_configs.py tests/test_slam_api.py -m gpu -q -x -k "track or Track" 2>&1 | tail -4 > gpurun_out/gpu_tests_track.log
tail -2 gpurun_out/gpu_tests_track.log
ab() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 full %.3f iterations %.3f' % (d['ms_per_step'], d['ms_per_step_iterations']))"; }
for k in 1 2 3 4; do
  LK_TRACK_COMP_INLINE=0 LK_TRACK_LOSS_INLINE=0 ab both_off
  LK_TRACK_COMP_INLINE=0 ab loss_inline
  LK_TRACK_LOSS_INLINE=0 ab comp_inline
  ab both_inline
done | tee gpurun_out/ab_track_loss.txt
bash tools/gpu_trace_modes.sh r4 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r4.md
sed -n 9,16p gpurun_out/trace_r4.md
