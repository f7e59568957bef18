#!/bin/bash
# round-6 closing job: GPU suite, bench + the three rocprofv3 passes, per-iteration timelines (overlapped and single-stream), per-stage counter traffic,
# SQ counters, the tracking chain (probe build), the measurement grid, the three end-to-end throughput runs
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 2>&1 | tail -8 ) > gpurun_out/gpu_tests_r6.log 2>&1
tail -5 gpurun_out/gpu_tests_r6.log
bash tools/profile_round.sh r6 2>&1 | tail -4
bash tools/gpu_trace_modes.sh r6 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r6.md
for mode in track geo color; do python tools/trace_summary.py /tmp/trace_$mode "$mode" gantt | sed -n '/^| start/,$p' > gpurun_out/gantt_r6_$mode.md; done
rm -rf /tmp/trace_cs
LK_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_cs -o t -- python tools/mode_trace.py color 40 > /tmp/trace_cs.log 2>&1
python tools/trace_summary.py /tmp/trace_cs "color, ONE stream (LK_SERIAL=1): every kernel alone on the chip" gantt 2>/dev/null | grep -v "only in" > gpurun_out/gantt_r6_color_serial.md
bash tools/stage_traffic.sh r6 > /dev/null 2>&1; grep "Per iteration" gpurun_out/stage_traffic_r6.md
bash tools/profile_sq.sh r6 > /dev/null 2>&1
cp loopy_slam_amd/libloopyhip.so /tmp/lib_ship.so
cp ab/lib_chain.so loopy_slam_amd/libloopyhip.so
timeout 300 python tools/probe/track_chain.py 40 > gpurun_out/track_chain_r6.md 2> gpurun_out/track_chain_r6.err
cp /tmp/lib_ship.so loopy_slam_amd/libloopyhip.so
grep -A7 "four launches" gpurun_out/track_chain_r6.md
timeout 900 python tools/sweep.py --md gpurun_out/sweep_r6.md > /dev/null 2> gpurun_out/sweep_r6.err; tail -2 gpurun_out/sweep_r6.err
timeout 300 python tools/slam_run.py --frames 51 --out gpurun_out/slam_run_room.json > /dev/null 2> gpurun_out/slam_run_room.err
for c in ScanNet/scene0000 TUM_RGBD/freiburg1_desk; do
  n=$(basename $c)
  timeout 400 python tools/slam_run.py --frames 31 --config configs/$c.yaml --out gpurun_out/slam_run_$n.json > /dev/null 2> gpurun_out/slam_run_$n.err
done
for f in gpurun_out/slam_run_room.json gpurun_out/slam_run_scene0000.json gpurun_out/slam_run_freiburg1_desk.json; do python -c "
import json; d = json.load(open('$f')); print('$f', 'tracked', d['ms_tracked_frame'], 'mapped steady', d.get('ms_mapped_frame_steady'), 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"; done
