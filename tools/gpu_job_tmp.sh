cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py 2> gpurun_out/bench_r3c.err | tail -1 > gpurun_out/bench_r3c.json
python -c "
import json; d = json.load(open('gpurun_out/bench_r3c.json')); print('%.2f ms/step, full %.2f ms/step' % (d['ms_per_step'], d['ms_per_step_full']), d['full_step']); print(d['config']['workload']); print(d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']); print(d['roofline']['kernel'], d['roofline']['frac'])"
tail -5 gpurun_out/bench_r3c.err
