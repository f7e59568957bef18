#!/bin/bash
# round 5, job B: the at-size / teacher-forced / workload / bench-CLI tests again (bounds set from job A's measurements), launch-by-launch
# timelines of the three iteration types with the split step
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loops_at_size.py tests/test_teacher_forced.py tests/test_workload.py "tests/test_bench_cli.py::test_one_rank_over_rccl_exchange_overhead" -m gpu -q 2>&1 | tail -150 > gpurun_out/r5b_tests.log
tail -12 gpurun_out/r5b_tests.log
bash tools/gpu_trace_modes.sh r5 > /dev/null 2>&1; grep -E "^period|host enqueue" gpurun_out/trace_r5.md
for mode in track geo color; do python tools/trace_summary.py /tmp/trace_$mode "$mode" gantt | sed -n '/^| start/,$p' > gpurun_out/gantt_r5_$mode.md; done
cat gpurun_out/gantt_r5_color.md
