cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
LOOPY_DIST_ONE_DEVICE=1 LOOPY_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/b2.err | tail -1 > gpurun_out/bench_r3_2rank_onedevice.json
grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*" gpurun_out/b2.err | tail -12
head -c 300 gpurun_out/bench_r3_2rank_onedevice.json; echo
for c in ; do
  n=$(basename $c)
  timeout 600 python tools/slam_run.py --frames 31 --config configs/$c.yaml --out gpurun_out/slam_run_$n.json > /dev/null 2> gpurun_out/slam_run_$n.err
  python -c "
import json; d = json.load(open('gpurun_out/slam_run_$n.json')); print('$n', 'tracked', d['ms_tracked_frame'], 'mapped', d['ms_mapped_frame'], 'mapped steady', d['ms_mapped_frame_steady'], 'fps', d['frames_per_s'], 'ate cm', d['ate_rmse_cm'])"
done
