#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/profile_round.sh r2 2>&1 | tail -6
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/prof_r2_sq
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_r2_sq -o b -- env LK_SERIAL=1 $B > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/prof_r2_sq > gpurun_out/pmc_sq_r2.txt 2>&1; head -30 gpurun_out/pmc_sq_r2.txt | cut -c1-220
find gpurun_out/prof_r2_sq -type f ! -name "b_counter_collection.csv" -delete
bash tools/gpu_trace_modes.sh r2f > /dev/null 2>&1
grep -E "^period|^###" gpurun_out/trace_r2f.md
du -sh gpurun_out/prof_r2_*
