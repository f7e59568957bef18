#!/bin/bash
# round-6 job E: the tracking iteration as a dependent chain (probe build ab/lib_chain.so: wall-clock stamps), plus the shipped library's per-iteration timeline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship.so
cp ab/lib_chain.so loopy_slam_amd/libloopyhip.so
timeout 300 python tools/probe/track_chain.py 40 > gpurun_out/track_chain_r6e.md 2> gpurun_out/track_chain_r6e.err
cp /tmp/lib_ship.so loopy_slam_amd/libloopyhip.so
tail -3 gpurun_out/track_chain_r6e.err
cat gpurun_out/track_chain_r6e.md | cut -c1-170
rm -rf /tmp/trace_track
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_track -o t -- python tools/mode_trace.py track 40 > /tmp/trace_track.log 2>&1
python tools/trace_summary.py /tmp/trace_track "track (R = 1500)" 2>/dev/null | head -14
