#!/bin/bash
# round 4, job C: per-iteration kernel timelines (rocprofv3 --kernel-trace) of the three iteration types with the 16 x 16 x 32 decoder forward
# on and off (LK_C16=0), same box; then alternating bench pairs.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
LK_C16=0 bash tools/gpu_trace_modes.sh c16off > /dev/null 2>&1
LK_C16=1 bash tools/gpu_trace_modes.sh c16on > /dev/null 2>&1
for t in c16off c16on; do echo "== $t"; grep -E "^period|k_decode_fwd|k_relpos_decode_fwd|k_relpos_fwd \|" gpurun_out/trace_$t.md; done
: > gpurun_out/ab_c16_fwd2.txt
for k in 1 2 3; do for v in 0 1; do
  LK_C16=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('LK_C16=$v  %.2f ms/step (full) %.2f (iterations)  decode_fwd %.3f' % (d['ms_per_step'], d['ms_per_step_iterations'], k.get('k_decode_fwd', 0)))" | tee -a gpurun_out/ab_c16_fwd2.txt
done; done
