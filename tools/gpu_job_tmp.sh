cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp loopy_slam_amd/libloopyhip.so /tmp/keep.so
timeout 900 python -m pytest tests/test_forward_parity.py tests/test_parity_at_size_configs.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in base3 knn2; do
cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
for c in ScanNet/scene0000 TUM_RGBD/freiburg1_desk; do
  n=$(basename $c)
  timeout 600 python tools/slam_run.py --frames 21 --config configs/$c.yaml --out gpurun_out/tmp_$n.json > /dev/null 2>&1
  python -c "
import json; d = json.load(open('gpurun_out/tmp_$n.json')); print('$v $n', 'tracked', d['ms_tracked_frame'], 'mapped steady', d['ms_mapped_frame_steady'], 'fps', d['frames_per_s'])"
done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v replica bench %.2f ms/step' % d['ms_per_step'])"
done
cp /tmp/keep.so loopy_slam_amd/libloopyhip.so
