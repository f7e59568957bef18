cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp loopy_slam_amd/libloopyhip.so /tmp/keep.so
timeout 900 python -m pytest tests/test_parity_at_size_configs.py tests/test_steps_parity.py tests/test_slam_api.py -m gpu -x -q 2>&1 | tail -3
python -c "
import json; d=json.load(open('gpurun_out/parity_at_size.json'))
for k,v in d.items():
    if 'scale' in k: print(k, v)"
for v in th16 xh16; do
cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
timeout 600 python tools/slam_run.py --frames 21 --config configs/ScanNet/scene0000.yaml --out gpurun_out/tmp_sc.json > /dev/null 2>&1
python -c "
import json; d = json.load(open('gpurun_out/tmp_sc.json')); print('$v scannet tracked', d['ms_tracked_frame'], 'mapped steady', d['ms_mapped_frame_steady'], 'ate', d['ate_rmse_cm'], 'l1', d['depth_l1_cm'])"
done
cp /tmp/keep.so loopy_slam_amd/libloopyhip.so
