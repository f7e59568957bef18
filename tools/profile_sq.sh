#!/bin/bash
# tools/profile_sq.sh TAG  (on the GPU box): SQ instruction / wait counters per kernel -> gpurun_out/sq_TAG.txt
tag=${1:-r2}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LK_SERIAL=1
mkdir -p gpurun_out
rm -rf /tmp/sq_$tag
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/sq_$tag -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /tmp/sq_$tag.log 2>&1
python tools/pmc_summary.py /tmp/sq_$tag > gpurun_out/sq_$tag.txt
head -30 gpurun_out/sq_$tag.txt
