cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp loopy_slam_amd/libloopyhip.so /tmp/keep.so
timeout 600 python -m pytest tests/test_steps_parity.py tests/test_slam_api.py -m gpu -x -q -k "track or slam_runs or exposure" 2>&1 | tail -3
for v in live fuse0 fuse1 live fuse1; do
cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
python tools/mode_trace.py track 40 --repeat 4 2>/dev/null | tail -2 | sed "s/^/$v /"
done
cp /tmp/keep.so loopy_slam_amd/libloopyhip.so
