set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 ./tools/probe/store_hazard_probe > gpurun_out/store_hazard_probe.txt 2>&1; tail -5 gpurun_out/store_hazard_probe.txt
timeout 600 python tools/probe/dh_store_insitu.py 40000 > gpurun_out/dh_store_insitu.txt 2>&1; cat gpurun_out/dh_store_insitu.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r2_base.json 2> gpurun_out/bench_r2_base.err; tail -c 400 gpurun_out/bench_r2_base.json
timeout 900 bash tools/gpu_trace_modes.sh r2base > /dev/null 2>&1; tail -30 gpurun_out/trace_r2base.md
timeout 1500 python -m pytest tests/test_parity_at_size.py -x -q 2>&1 | tail -15 > gpurun_out/parity_at_size.log; cat gpurun_out/parity_at_size.log
