#!/bin/bash
# round 4, job B: the 16 x 16 x 32 decoder forward (k_decode_fwd16) against the 32 x 32 x 16 kernels on ONE box (LK_C16=0 switches back):
# the GPU suite with the new form, then alternating bench runs with the per-kernel table of the profiled step.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
: > gpurun_out/ab_c16_fwd.txt
for k in 1 2 3; do for v in 0 1; do
  LK_C16=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('LK_C16=$v  %.2f ms/step (full) %.2f (iterations)  decode_fwd %.3f  relpos_fwd %.3f decode_bwd %.3f' % (d['ms_per_step'], d['ms_per_step_iterations'], k.get('k_decode_fwd', 0), k.get('k_relpos_fwd', 0), k.get('k_decode_bwd', 0)))" | tee -a gpurun_out/ab_c16_fwd.txt
done; done
for v in 0 1; do
  LK_C16=$v LK_SERIAL=1 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('LK_C16=$v serial %.2f ms/step: ' % d['ms_per_step_iterations'] + '  '.join('%s=%.2f' % (n[2:], x) for n, x in list(k.items())[:10]))" | tee -a gpurun_out/ab_c16_fwd.txt
done
python -c "
import sys; sys.path.insert(0, '.')
from loopy_slam_amd import core
import ctypes
e = core.Engine(); out = (ctypes.c_int32 * 5)(); e.lib.dll.lk_debug_occupancy(out); print('occupancy (decode_fwd, decode_bwd, relpos_fwd, relpos_bwd_fused, wgrad):', list(out))" | tee -a gpurun_out/ab_c16_fwd.txt
