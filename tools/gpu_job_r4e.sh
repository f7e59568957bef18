#!/bin/bash
# round 4, job E: GPU suite (overlapped exchange, accuracy vs the oracle fixtures, both decoder forms), full-frame render beside the tracking
# (A/B LOOPY_RENDER_INLINE), the exchange machinery with one rank over RCCL (A/B LOOPY_DIST_OVERLAP)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
: > gpurun_out/ab_render_stream.txt
for k in 1 2 3; do for v in 1 0; do
  LOOPY_RENDER_INLINE=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('LOOPY_RENDER_INLINE=$v  %.2f ms/step (full) %.2f (iterations only)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/ab_render_stream.txt
done; done
: > gpurun_out/ab_dist1.txt
for k in 1 2; do for v in 1 0; do
  LOOPY_DIST_FORCE=1 LOOPY_DIST_OVERLAP=$v python bench.py --no-cpu-baseline 2>gpurun_out/dist1.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('1 rank over RCCL, LOOPY_DIST_OVERLAP=$v  %.2f ms/step (full) %.2f (iterations only)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/ab_dist1.txt
done; done
