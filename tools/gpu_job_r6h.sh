#!/bin/bash
# round-6 job H: tracking-chain changes (two candidate rounds + early loads in the search, the loss prologue's loads in flight together, L2 warm-up):
# parity tests, chain stamps (probe build), alternating bench runs against the library before them
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_steps_parity.py tests/test_forward_parity.py tests/test_parity_at_size.py tests/test_loops_at_size.py tests/test_fullsize_gpu.py tests/test_teacher_forced.py tests/test_slam_api.py -m gpu -q 2>&1 | tail -6 ) 2>&1
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship.so
cp ab/lib_chain.so loopy_slam_amd/libloopyhip.so
timeout 300 python tools/probe/track_chain.py 40 > gpurun_out/track_chain_r6h.md 2> gpurun_out/track_chain_r6h.err
cp /tmp/lib_ship.so loopy_slam_amd/libloopyhip.so
tail -3 gpurun_out/track_chain_r6h.err
grep -A7 "four launches" gpurun_out/track_chain_r6h.md
rm -rf /tmp/trace_track
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_track -o t -- python tools/mode_trace.py track 40 > /tmp/trace_track.log 2>&1
python tools/trace_summary.py /tmp/trace_track "track (R = 1500)" 2>/dev/null | head -12
bash tools/ab_quick.sh 3 base new
cp /tmp/lib_ship.so loopy_slam_amd/libloopyhip.so
