#!/bin/bash
# scratch GPU job (gpurun -- 'bash tools/gpu_job.sh'): full GPU test suite, smoke, then the bench line
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err
tail -c 700 gpurun_out/bench_r2.json
