#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
lscpu | grep "Model name" | head -1
( time timeout 1500 python -m pytest tests/test_parity_at_size.py tests/test_parity_at_size_configs.py tests/test_loops_at_size.py tests/test_teacher_forced.py tests/test_workload.py tests/test_forward_parity.py tests/test_backward_parity.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | head -40 ) 2>&1 | tee gpurun_out/r5j_tests.log | tail -30
