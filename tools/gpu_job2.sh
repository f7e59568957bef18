#!/bin/bash
# scratch: quick bench (overlapped x3, serial) + colour-iteration fork timeline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
for k in 1 2 3; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('overlap %.2f ms/step' % d['ms_per_step'], {k[2:]: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
done
LK_SERIAL=1 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial %.2f ms/step' % d['ms_per_step'], {k[2:]: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
rm -rf /tmp/trc; rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -o t -- python tools/mode_trace.py color 40 > /dev/null 2>&1
python tools/trace_fork.py /tmp/trc
