cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r2 2>&1 | tail -12
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_r2_sq -o b -- env LK_SERIAL=1 $B > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/prof_r2_sq > gpurun_out/pmc_sq_r2.txt 2>&1; cat gpurun_out/pmc_sq_r2.txt
find gpurun_out/prof_r2_sq -type f ! -name "b_counter_collection.csv" -delete; du -sh gpurun_out/prof_r2_*
