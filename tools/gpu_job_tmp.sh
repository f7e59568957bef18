cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp loopy_slam_amd/libloopyhip.so /tmp/keep.so
timeout 900 python -m pytest tests/test_parity_at_size.py tests/test_steps_parity.py tests/test_slam_api.py -m gpu -x -q 2>&1 | tail -2
python -c "
import json; d=json.load(open('gpurun_out/parity_at_size.json'))
for k,v in d.items():
    if k.startswith('track'): print(k, {a:('%.2e'%b if isinstance(b,float) else b) for a,b in v.items() if a.startswith('g[')})"
for v in knn2 th16 knn2 th16; do
cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
python tools/mode_trace.py track 40 --repeat 3 2>/dev/null | tail -1 | sed "s/^/$v /"
done
for v in knn2 th16; do
cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
timeout 600 python tools/slam_run.py --frames 21 --config configs/TUM_RGBD/freiburg1_desk.yaml --out gpurun_out/tmp_tum.json > /dev/null 2>&1
python -c "
import json; d = json.load(open('gpurun_out/tmp_tum.json')); print('$v tum tracked', d['ms_tracked_frame'], 'mapped steady', d['ms_mapped_frame_steady'], 'ate', d['ate_rmse_cm'])"
done
cp /tmp/keep.so loopy_slam_amd/libloopyhip.so
