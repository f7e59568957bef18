#!/bin/bash
# round-3 GPU job b: parity tests (incl. the TUM / ScanNet at-size cases), ScanNet / TUM end-to-end runs with exposure in the native loops
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
for c in ScanNet/scene0000 TUM_RGBD/freiburg1_desk; do
  n=$(basename $c)
  timeout 600 python tools/slam_run.py --frames 31 --config configs/$c.yaml --out gpurun_out/slam_run_$n.json > /dev/null 2> gpurun_out/slam_run_$n.err
  python -c "
import json; d = json.load(open('gpurun_out/slam_run_$n.json')); print('$n', 'tracked', d['ms_tracked_frame'], 'mapped', d['ms_mapped_frame'], 'mapped steady', d['ms_mapped_frame_steady'], 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"
done
