#!/bin/bash
# round-4 closing job: GPU suite, bench + the three rocprofv3 passes, per-iteration timelines, SQ counters, the accuracy runs of the product
# on the TUM / ScanNet configs (the oracle's runs are fixtures made in the build container), the three end-to-end throughput runs.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/gpu_tests.log
tail -2 gpurun_out/gpu_tests.log
bash tools/profile_round.sh r4 2>&1 | tail -4
bash tools/gpu_trace_modes.sh r4 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r4.md
for mode in track geo color; do python tools/trace_summary.py /tmp/trace_$mode "$mode" gantt | sed -n '/^| start/,$p' > gpurun_out/gantt_$mode.md; done
bash tools/profile_sq.sh r4 > /dev/null 2>&1
# mapping.fix_geo_decoder: False inside the native loop: per-iteration timelines with the geometry decoder trained (k_geo_wgrad timed on the chip)
for mode in geo color; do
  rm -rf /tmp/trace_gf_$mode
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_gf_$mode -o t -- python tools/mode_trace.py $mode 40 --geo-free > /tmp/trace_gf_$mode.log 2>&1
  python tools/trace_summary.py /tmp/trace_gf_$mode "$mode, fix_geo_decoder: False (R = 5000, N = 100 000)" >> gpurun_out/trace_r4_geofree.md
done
grep -E "^period|k_geo_wgrad" gpurun_out/trace_r4_geofree.md
for s in 1219 1220 1221 1222 1223; do
  timeout 300 python tools/accuracy_run.py --pipeline hip --config configs/ScanNet/scene0000.yaml --frames 50 --rays 2000 --color-refine 0 --seed $s --out gpurun_out/acc_scannet_hip_s$s.json 2> gpurun_out/acc_scannet_hip_s$s.err | cut -c1-200
  timeout 300 python tools/accuracy_run.py --pipeline hip --config configs/TUM_RGBD/freiburg1_desk.yaml --frames 50 --rays 2000 --iters-scale 0.5 --color-refine 0 --seed $s --out gpurun_out/acc_tum_hip_s$s.json 2> gpurun_out/acc_tum_hip_s$s.err | cut -c1-200
done
rm -f gpurun_out/acc_room_hip_fullrays.json
timeout 300 python tools/slam_run.py --frames 51 --out gpurun_out/slam_run_room.json > /dev/null 2> gpurun_out/slam_run_room.err
for c in ScanNet/scene0000 TUM_RGBD/freiburg1_desk; do
  n=$(basename $c)
  timeout 400 python tools/slam_run.py --frames 31 --config configs/$c.yaml --out gpurun_out/slam_run_$n.json > /dev/null 2> gpurun_out/slam_run_$n.err
done
for f in gpurun_out/slam_run_*.json; do python -c "
import json; d = json.load(open('$f')); print('$f', 'tracked', d['ms_tracked_frame'], 'mapped steady', d.get('ms_mapped_frame_steady'), 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"; done
