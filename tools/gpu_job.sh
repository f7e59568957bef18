cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/mode_trace.py geo 40 --repeat 6 2>/dev/null
python tools/mode_trace.py track 40 --repeat 4 2>/dev/null
AMD_LOG_LEVEL=0 HIP_LAUNCH_BLOCKING=0 python tools/host_overhead.py 2>/dev/null | tail -4
