#!/bin/bash
# round 4, job F: the library's own RCCL communicator on the launch stream (loopy_slam_amd/rccl.py) against torch's collectives, one rank
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gloo.py -m gpu -q -x 2>&1 | tail -5
: > gpurun_out/ab_dist_direct.txt
for k in 1 2 3; do for v in plain torch direct; do
  if [ $v = plain ]; then env="X=1"; elif [ $v = torch ]; then env="LOOPY_DIST_FORCE=1 LOOPY_DIST_TORCH=1"; else env="LOOPY_DIST_FORCE=1"; fi
  env $env python bench.py --no-cpu-baseline 2>gpurun_out/dist_direct.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v  %.2f ms/step (full) %.2f (iterations only)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/ab_dist_direct.txt
done; done
tail -3 gpurun_out/dist_direct.err
