#!/bin/bash
# round 5, job A: the new at-size / teacher-forced / bench-CLI / workload tests on the chip, then the split step (colour trunk's reduction + Adam +
# repack on the weight-gradient stream) against the variant without it (ab/lib_nosplit.so, -DLK_SPLIT_STEP=0): per-iteration wall times and
# alternating bench pairs on ONE box
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
cp ab/lib_split.so loopy_slam_amd/libloopyhip.so
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1300 python -m pytest tests/test_loops_at_size.py tests/test_teacher_forced.py tests/test_workload.py tests/test_bench_cli.py tests/test_dist_gloo.py tests/test_steps_parity.py -m gpu -q 2>&1 | tail -60 > gpurun_out/r5a_tests.log
tail -25 gpurun_out/r5a_tests.log
: > gpurun_out/r5a_ab_split.txt
for k in 1 2; do for v in split nosplit; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  for mode in color track geo; do
    echo "$v $(python tools/mode_trace.py $mode 40 2>/dev/null | tail -1)" | tee -a gpurun_out/r5a_ab_split.txt
  done
done; done
for k in 1 2 3; do for v in split nosplit; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v %.2f ms/step (full) %.2f (iterations)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/r5a_ab_split.txt
done; done
cp ab/lib_split.so loopy_slam_amd/libloopyhip.so
python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -c 1500 gpurun_out/r5a_bench.json
