cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python tools/sweep.py --md gpurun_out/r3_sweep.md > gpurun_out/sweep.log 2>&1
tail -45 gpurun_out/sweep.log
