#!/bin/bash
# round 5, job G: the rel-pos tile reduction of the launch stream's reduction launch at 4 / 8 / 16 / 32 elements per workgroup (ab/lib_rp*.so; lib_split = 32)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
cp ab/lib_rp8.so loopy_slam_amd/libloopyhip.so
timeout 600 python -m pytest tests/test_steps_parity.py tests/test_backward_parity.py tests/test_parity_at_size.py -m gpu -q -k "not tum" 2>&1 | tail -5
: > gpurun_out/r5g_ab_rp.txt
for k in 1 2 3; do for v in split rp16 rp8 rp4; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v %.2f ms/step (full) %.2f (iterations)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/r5g_ab_rp.txt
done; done
cp ab/lib_rp8.so loopy_slam_amd/libloopyhip.so
rm -rf /tmp/trace_color; rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_color -o t -- python tools/mode_trace.py color 40 > /tmp/trace_color.log 2>&1
python tools/trace_summary.py /tmp/trace_color "color" gantt | sed -n '/^| start/,$p' | tee gpurun_out/gantt_r5g_color.md
