#!/bin/bash
# tools/pmc_mem.sh TAG  (on the GPU box): memory-path counters (TA / TCP / UTCL1) per kernel, LK_SERIAL=1 bench steps.
tag=${1:-mem}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LK_SERIAL=1
mkdir -p gpurun_out
out=gpurun_out/pmc_${tag}.txt; : > $out
pass() {
  rm -rf /tmp/pmc_$1
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1 -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmc_$1.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_$1 | head -14 >> $out; echo >> $out
}
pass a "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE"
pass b "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS GRBM_GUI_ACTIVE"
# (a TCC_* pass - TCC_HIT TCC_MISS TCC_EA0_RDREQ ... - did not finish within 12 minutes on this pool: 16 channels x 8 XCDs of instances; left out)
pass d "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE"
cat $out
