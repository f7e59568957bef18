#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/gpu_tests.log
tail -2 gpurun_out/gpu_tests.log
for k in 1 2 3; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('overlap %.2f ms/step' % d['ms_per_step'])"
done
python tools/probe/step_phases.py 2>&1 | tail -2
rm -rf /tmp/trb
rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o t -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python tools/trace_gaps.py /tmp/trb 15 | head -14
