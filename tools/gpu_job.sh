#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/slam_run.py --frames 51 --out gpurun_out/r2_slam_run.json 2>&1 | tail -2 | cut -c1-700
timeout 600 python tools/slam_run.py --frames 31 --config configs/TUM_RGBD/freiburg1_desk.yaml --out gpurun_out/r2_slam_run_tum.json 2>&1 | tail -2 | cut -c1-700
timeout 600 python tools/slam_run.py --frames 31 --config configs/ScanNet/scene0000.yaml --out gpurun_out/r2_slam_run_scannet.json 2>&1 | tail -2 | cut -c1-700
