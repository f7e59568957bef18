#!/bin/bash
# round-5 closing job: GPU suite, bench + the three rocprofv3 passes, per-iteration timelines, SQ counters, the three end-to-end throughput runs
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/gpu_tests_r5.log 2>&1
tail -5 gpurun_out/gpu_tests_r5.log
bash tools/profile_round.sh r5 2>&1 | tail -4
bash tools/gpu_trace_modes.sh r5 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r5.md
for mode in track geo color; do python tools/trace_summary.py /tmp/trace_$mode "$mode" gantt | sed -n '/^| start/,$p' > gpurun_out/gantt_r5_$mode.md; done
bash tools/profile_sq.sh r5 > /dev/null 2>&1
timeout 300 python tools/slam_run.py --frames 51 --out gpurun_out/slam_run_room.json > /dev/null 2> gpurun_out/slam_run_room.err
for c in ScanNet/scene0000 TUM_RGBD/freiburg1_desk; do
  n=$(basename $c)
  timeout 400 python tools/slam_run.py --frames 31 --config configs/$c.yaml --out gpurun_out/slam_run_$n.json > /dev/null 2> gpurun_out/slam_run_$n.err
done
for f in gpurun_out/slam_run_room.json gpurun_out/slam_run_scene0000.json gpurun_out/slam_run_freiburg1_desk.json; do python -c "
import json; d = json.load(open('$f')); print('$f', 'tracked', d['ms_tracked_frame'], 'mapped steady', d.get('ms_mapped_frame_steady'), 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"; done
