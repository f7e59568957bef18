#!/bin/bash
# round 4, job D: shader-clock phase stamps of the decoder forward, 16 x 16 x 32 form and (LK_C16=0) the 32 x 32 x 16 form, same box
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
cp loopy_slam_amd/libloopyhip.so /tmp/lib_keep.so
cp ab/lib_clk.so loopy_slam_amd/libloopyhip.so
LK_C16=1 python tools/probe/decode_clock16.py > gpurun_out/decode_clock16.txt 2>&1
LK_C16=0 python tools/probe/decode_clock.py > gpurun_out/decode_clock32.txt 2>&1
cp /tmp/lib_keep.so loopy_slam_amd/libloopyhip.so
cat gpurun_out/decode_clock16.txt; echo; cat gpurun_out/decode_clock32.txt
