#!/bin/bash
# tools/stage_traffic.sh TAG [ITERS]  (on the GPU box): HBM-side traffic of ONE iteration type at a time - the bench's kernels are shared between
# the 'geometry' and 'color' iterations (k_decode_fwd, k_decode_bwd, k_feat_gather, k_bwd_reduce ...), so the per-kernel averages of a bench
# run blend small and large launches (round-5 review, "missing" 4).  Three rocprofv3 passes per mode over tools/mode_trace.py, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes for the TCC counters (FETCH_SIZE and WRITE_SIZE do not fit one pass): --kernel-trace only,
# --pmc FETCH_SIZE, --pmc WRITE_SIZE.  -> gpurun_out/stage_traffic_TAG.md (tools/stage_traffic.py)
tag=${1:-r6}
iters=${2:-40}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in ${MODES:-color geo track}; do
  for pass in time fetch write; do
    d=/tmp/st_${mode}_$pass
    rm -rf $d
    case $pass in
      time)  extra="--stats" ;;
      fetch) extra="--pmc FETCH_SIZE" ;;
      write) extra="--pmc WRITE_SIZE" ;;
    esac
    rocprofv3 --kernel-trace $extra --output-format csv -d $d -o t -- python tools/mode_trace.py $mode $iters --repeat 2 ${MODE_ARGS} > /tmp/st_${mode}_$pass.log 2>&1
  done
done
python tools/stage_traffic.py $tag $iters 3 ${MODES:-color geo track} > gpurun_out/stage_traffic_$tag.md
cat gpurun_out/stage_traffic_$tag.md
