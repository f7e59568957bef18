#!/bin/bash
# round 4, job I: the tracker's fused forward with the colour waves on a barrier of their own (LK_SOFTBAR=0 switches back), same box;
# k_geo_wgrad with up to 768 workgroups
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "track or tracker or slam or accuracy or fullsize" 2>&1 | tail -3
: > gpurun_out/ab_softbar.txt
for v in 0 1; do
  rm -rf /tmp/trace_sb$v
  LK_SOFTBAR=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_sb$v -o t -- python tools/mode_trace.py track 40 > /tmp/trace_sb$v.log 2>&1
  echo "== LK_SOFTBAR=$v" | tee -a gpurun_out/ab_softbar.txt
  python tools/trace_summary.py /tmp/trace_sb$v "track" | grep -E "^period|k_relpos_decode_fwd|k_relpos_interp_bwd|k_decode_bwd|k_sample_interp_pose" | tee -a gpurun_out/ab_softbar.txt
done
for k in 1 2 3; do for v in 0 1; do
  LK_SOFTBAR=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('LK_SOFTBAR=$v  %.2f ms/step (full) %.2f (iterations only)' % (d['ms_per_step'], d['ms_per_step_iterations']))" | tee -a gpurun_out/ab_softbar.txt
done; done
rm -rf /tmp/trace_gf; rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_gf -o t -- python tools/mode_trace.py geo 40 --geo-free > /tmp/trace_gf.log 2>&1
python tools/trace_summary.py /tmp/trace_gf "geo, fix_geo_decoder: False, 768 workgroups" | grep -E "^period|k_geo_wgrad|k_reduce_partials" | tee gpurun_out/geo_wgrad_768.txt
