#!/bin/bash
# round-6 job C: the default bench line with the two new workloads (wall clock), per-stage traffic tables with their JSON
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py --no-cpu-baseline > gpurun_out/bench_r6c.json 2> gpurun_out/bench_r6c.err ) 2>&1 | tail -3
tail -3 gpurun_out/bench_r6c.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r6c.json'))
print('ms_per_step', d['ms_per_step'], 'steps', d['steps'])
for k, v in d.get('workloads', {}).items():
    print(k, {q: (round(x, 3) if isinstance(x, float) else x) for q, x in v.items() if q != 'workload'})
PY
bash tools/stage_traffic.sh r6c 40 > /dev/null 2>&1
grep "Per iteration" gpurun_out/stage_traffic_r6c.md
ls -la gpurun_out/stage_traffic_r6c.*
