#!/bin/bash
# A/B timing of library variants on one box: tools/gpu_job_ab.sh NAME...
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
cp loopy_slam_amd/libloopyhip.so /tmp/lib_keep.so
bash tools/ab_run.sh "$@"
bash tools/ab_quick.sh 2 "$@"
cp /tmp/lib_keep.so loopy_slam_amd/libloopyhip.so
