#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/ab_dist_direct2.txt
for k in 1 2; do for v in plain torch torch_overlap direct direct_overlap; do
  case $v in
    plain) env="X=1";;
    torch) env="LOOPY_DIST_FORCE=1 LOOPY_DIST_TORCH=1";;
    torch_overlap) env="LOOPY_DIST_FORCE=1 LOOPY_DIST_TORCH=1 LOOPY_DIST_OVERLAP=1";;
    direct) env="LOOPY_DIST_FORCE=1";;
    direct_overlap) env="LOOPY_DIST_FORCE=1 LOOPY_DIST_OVERLAP=1";;
  esac
  env $env python bench.py --no-cpu-baseline 2>gpurun_out/dist_direct.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v  %.2f ms/step (full) %.2f (iterations only)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/ab_dist_direct2.txt
done; done
