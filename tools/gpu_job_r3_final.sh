#!/bin/bash
# round-3 closing job: tests, bench + the three rocprofv3 passes, per-iteration timelines, SQ counters, the three end-to-end runs,
# the 2-rank functional bench line.  Everything lands under gpurun_out/; tools/summarize_prof.py etc. copy the summaries to profiles/.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/gpu_tests.log
tail -2 gpurun_out/gpu_tests.log
bash tools/profile_round.sh r3 2>&1 | tail -4
bash tools/gpu_trace_modes.sh r3 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r3.md
bash tools/profile_sq.sh r3 > /dev/null 2>&1
timeout 300 python tools/slam_run.py --frames 51 --out gpurun_out/slam_run_room.json > /dev/null 2> gpurun_out/slam_run_room.err
for c in ScanNet/scene0000 TUM_RGBD/freiburg1_desk; do
  n=$(basename $c)
  timeout 400 python tools/slam_run.py --frames 31 --config configs/$c.yaml --out gpurun_out/slam_run_$n.json > /dev/null 2> gpurun_out/slam_run_$n.err
done
for f in gpurun_out/slam_run_*.json; do python -c "
import json; d = json.load(open('$f')); print('$f', 'tracked', d['ms_tracked_frame'], 'mapped steady', d.get('ms_mapped_frame_steady'), 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"; done
LOOPY_DIST_ONE_DEVICE=1 LOOPY_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_r3_2rank.err | tail -1 > gpurun_out/bench_r3_2rank.json
python -c "
import json; d = json.load(open('gpurun_out/bench_r3_2rank.json')); print('2 ranks / 1 device (gloo, host-staged): n_gpus', d['n_gpus'], '%.2f ms/step' % d['ms_per_step'], d['scaling'])"
