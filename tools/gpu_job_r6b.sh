#!/bin/bash
# round-6 job B: unorm16 derivative mask + range guard: parity tests, bench, colour-stage traffic
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_operand_range.py tests/test_backward_parity.py tests/test_forward_parity.py tests/test_parity_at_size.py tests/test_parity_at_size_configs.py tests/test_split_step_order.py tests/test_loops_at_size.py tests/test_steps_parity.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r6b_tests.log 2>&1
tail -8 gpurun_out/r6b_tests.log
for i in 1 2; do
timeout 300 python bench.py --headline-only --no-cpu-baseline > gpurun_out/bench_r6b_$i.json 2> gpurun_out/bench_r6b.err
python - <<PY
import json
d = json.load(open('gpurun_out/bench_r6b_$i.json'))
print('ms_per_step', d['ms_per_step'], 'iterations', d.get('ms_per_step_iterations'), 'roofline', d['roofline']['kernel'], d['roofline']['avg_launch_us'], d.get('kernel_ms_per_step'))
PY
done
MODES=color bash tools/stage_traffic.sh r6b 40 > /dev/null 2>&1
sed -n 8,24p gpurun_out/stage_traffic_r6b.md | cut -c1-160; tail -2 gpurun_out/stage_traffic_r6b.md
