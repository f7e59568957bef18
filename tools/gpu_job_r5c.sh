#!/bin/bash
# round 5, job C: the at-size loop tests and the teacher-forced replays with the perturbed-oracle yardstick
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loops_at_size.py tests/test_teacher_forced.py tests/test_workload.py -m gpu -q 2>&1 | tail -150 > gpurun_out/r5c_tests.log
tail -12 gpurun_out/r5c_tests.log
