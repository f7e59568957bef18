#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
lscpu | grep "Model name" | head -1
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | head -40 ) 2>&1 | tee gpurun_out/r5k_suite.log | tail -20
