#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python tools/probe/split_step_race.py 6 2>&1 | grep -v "^$" | tee gpurun_out/r5e_split_race.txt | tail -40
cp ab/lib_split.so loopy_slam_amd/libloopyhip.so
