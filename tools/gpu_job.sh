#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
for a in "--warmup 1 --steps 3" "--warmup 15 --steps 20" "--warmup 1 --steps 20" "--warmup 15 --steps 3" "--warmup 1 --steps 3"; do
timeout 900 python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$a: %.2f ms/step' % d['ms_per_step'])"
done
