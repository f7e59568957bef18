#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/gpu_tests.log
tail -4 gpurun_out/gpu_tests.log
for k in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('overlap %.2f ms/step' % d['ms_per_step'], {k[2:]: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
done
LK_SERIAL=1 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial %.2f ms/step' % d['ms_per_step'], {k[2:]: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})"
M="python tools/mode_trace.py color 10 --repeat 1"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  LK_SERIAL=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$tag -o b -- $M > /tmp/pmc_$tag.log 2>&1
  echo "== $set"; python tools/pmc_summary.py /tmp/pmc_$tag 2>&1 | grep -E "^kernel|relpos|decode_bwd|wgrad"
done > gpurun_out/pmc_fused.txt 2>&1
cat gpurun_out/pmc_fused.txt | cut -c1-250
