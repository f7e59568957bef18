#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_at_size.py -m gpu -q -k "ba_mode or tracker_iteration" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -40
