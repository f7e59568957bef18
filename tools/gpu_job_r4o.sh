#!/bin/bash
# k_dw2_hbar as a rider of k_feat_gather: tests, same-box A/B, launch-by-launch timeline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_steps_parity.py tests/test_parity_at_size.py tests/test_backward_parity.py tests/test_dist_gloo.py -m gpu -q -x 2>&1 | grep -E "^E  |^FAILED|passed|failed" | head -6
ab() { timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 full %.3f iterations %.3f' % (d['ms_per_step'], d['ms_per_step_iterations']))"; }
for k in 1 2 3 4; do
  LK_DW2_RIDER=0 ab own_launch
  ab rider
done | tee gpurun_out/ab_dw2_rider.txt
rm -rf /tmp/trace_color
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_color -o t -- python tools/mode_trace.py color 40 > /tmp/trace_color.log 2>&1
python tools/trace_summary.py /tmp/trace_color "color" gantt > gpurun_out/gantt_color_dw2.md
grep -E "^period" gpurun_out/gantt_color_dw2.md; sed -n '/^| start/,$p' gpurun_out/gantt_color_dw2.md
