#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/gpu_tests.log
tail -2 gpurun_out/gpu_tests.log
for k in 1 2 3; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('overlap %.2f ms/step' % d['ms_per_step'], {k[2:]: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
done
for m in geo color track; do python tools/mode_trace.py $m 40 --repeat 3 2>&1 | tail -1; done
