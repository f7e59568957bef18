#!/bin/bash
# tools/pmc_modes.sh MODE TAG  (on the GPU box): SQ / instruction-cache counter passes over one iteration type (LK_SERIAL=1).
mode=${1:-track}; tag=${2:-pmc}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LK_SERIAL=1
mkdir -p gpurun_out
out=gpurun_out/pmc_${tag}_$mode.txt; : > $out
pass() {
  rm -rf /tmp/pmc_$1
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1 -o b -- python tools/mode_trace.py $mode 10 --repeat 1 > /tmp/pmc_$1.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_$1 >> $out; echo >> $out
}
pass a "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
pass b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"
pass c "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES"
cat $out
