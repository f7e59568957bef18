#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_loops_at_size.py::test_map_call_at_bench_size tests/test_teacher_forced.py::test_config1_ten_frames_teacher_forced -m gpu -q 2>&1 | tail -60 > gpurun_out/r5d_tests.log
tail -8 gpurun_out/r5d_tests.log
