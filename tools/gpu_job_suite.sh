#!/bin/bash
# the whole -m gpu suite as the driver runs it (plus durations): gpurun_out/gpu_suite.log
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q --maxfail=6 --durations=15 2>&1 | tail -40 ) > gpurun_out/gpu_suite.log 2>&1
tail -30 gpurun_out/gpu_suite.log
