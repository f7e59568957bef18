#!/bin/bash
# round 4, first job: the GPU suite (float64 referee, run.py, load_pretrain), the bench line with the full step as headline, and the
# accuracy runs of the product on the config-1 sequence (three seeds) + the default Replica budget.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
timeout 600 python bench.py 2> gpurun_out/bench_r4a.err | tail -1 > gpurun_out/bench_r4a.json
python -c "
import json; d = json.load(open('gpurun_out/bench_r4a.json')); print('bench: %.2f ms/step (full), %.2f iterations only, %.2f M rays/s' % (d['ms_per_step'], d['ms_per_step_iterations'], d['value'] / 1e6), d['roofline']['kernel'], d['roofline']['frac'])"
for s in 1219 1220 1221; do
  timeout 300 python tools/accuracy_run.py --pipeline hip --config configs/Synthetic/room.yaml --frames 50 --rays 500 --color-refine 0 --seed $s --out gpurun_out/acc_room_hip_s$s.json 2> gpurun_out/acc_room_hip_s$s.err | cut -c1-400
done
timeout 300 python tools/accuracy_run.py --pipeline hip --config configs/Synthetic/room.yaml --frames 50 --rays 0 --color-refine 0 --out gpurun_out/acc_room_hip_fullrays.json 2> gpurun_out/acc_room_hip_fullrays.err | cut -c1-400
