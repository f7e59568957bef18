#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/probe/track_grad_outliers.py replica 1500 2>&1 | grep -v "^$" | tee gpurun_out/r5i_outliers.txt | tail -40
