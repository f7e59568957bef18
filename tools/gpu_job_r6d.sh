#!/bin/bash
# round-6 job D: wgrad tile reduction - parity tests, A/B bench against the previous library (ab/lib_prev.so if present), colour traffic
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_backward_parity.py tests/test_parity_at_size.py tests/test_loops_at_size.py tests/test_split_step_order.py tests/test_bench_cli.py tests/test_slam_api.py -m gpu -q -x 2>&1 | tail -6 ) 2>&1
if [ -f ab/lib_prev.so ]; then cp loopy_slam_amd/libloopyhip.so ab/lib_new.so; bash tools/ab_quick.sh 3 prev new; cp ab/lib_new.so loopy_slam_amd/libloopyhip.so; fi
MODES=color bash tools/stage_traffic.sh r6d 40 > /dev/null 2>&1
sed -n 8,16p gpurun_out/stage_traffic_r6d.md | cut -c1-140; grep "Per iteration" gpurun_out/stage_traffic_r6d.md
