#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/gpu_tests.log
tail -2 gpurun_out/gpu_tests.log
bash tools/profile_round.sh r4 2>&1 | tail -2
bash tools/gpu_trace_modes.sh r4 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r4.md
timeout 300 python tools/slam_run.py --frames 51 --out gpurun_out/slam_run_room.json > /dev/null 2> gpurun_out/slam_run_room.err
python -c "
import json; d = json.load(open('gpurun_out/slam_run_room.json')); print('slam_run room: tracked', d['ms_tracked_frame'], 'mapped steady', d.get('ms_mapped_frame_steady'), 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"
