#!/bin/bash
# round-3 GPU job: parity tests, the 1-GPU bench line, and the 2-ranks-on-one-device functional run of the multi-GPU bench path
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/gpu_tests.log
tail -2 gpurun_out/gpu_tests.log
python bench.py --no-cpu-baseline 2> gpurun_out/bench_r3.err | tail -1 > gpurun_out/bench_r3.json
python -c "
import json; d = json.load(open('gpurun_out/bench_r3.json')); print('1 GPU %.2f ms/step' % d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'], 4), {k[2:]: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
LOOPY_DIST_ONE_DEVICE=1 LOOPY_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_r3_2rank.err | tail -1 > gpurun_out/bench_r3_2rank.json
python -c "
import json; d = json.load(open('gpurun_out/bench_r3_2rank.json')); print('2 ranks / 1 device (gloo, host-staged): n_gpus', d['n_gpus'], '%.2f ms/step' % d['ms_per_step'], d['scaling'])"
tail -3 gpurun_out/bench_r3_2rank.err
