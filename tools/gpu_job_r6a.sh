#!/bin/bash
# round-6 job A: the split-step ordering tests + the loops at size, the bench line as the round starts, per-stage traffic tables
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_split_step_order.py tests/test_loops_at_size.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r6a_tests.log 2>&1
tail -8 gpurun_out/r6a_tests.log
timeout 300 python bench.py --headline-only > gpurun_out/bench_r6a.json 2> gpurun_out/bench_r6a.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r6a.json'))
print('ms_per_step', d['ms_per_step'], 'iterations', d.get('ms_per_step_iterations'), 'roofline', d['roofline']['kernel'], d['roofline']['avg_launch_us'])
PY
bash tools/stage_traffic.sh r6a 40 > /dev/null 2>&1
tail -60 gpurun_out/stage_traffic_r6a.md
