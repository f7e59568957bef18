#!/bin/bash
# tools/profile_round.sh TAG   (on the GPU box): bench JSON + the three rocprofv3 passes the summaries are made from.
# Outputs under gpurun_out/: bench_TAG.json, prof_TAG_{stats,fetch,write}/ ; then `python tools/summarize_prof.py TAG` here.
tag=${1:-r1}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -c 600 gpurun_out/bench_$tag.json; echo
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only"
rm -rf gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -o b -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -o b -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -o b -- $B > /dev/null 2>&1
find gpurun_out/prof_${tag}_* -name "*.csv" | head -20
# keep only what the summaries read (the traces are large)
find gpurun_out/prof_${tag}_* -type f ! -name "b_kernel_stats.csv" ! -name "b_counter_collection.csv" -delete
du -sh gpurun_out/prof_${tag}_*
