#!/bin/bash
# tools/gpu_trace_modes.sh TAG  (on the GPU box): rocprofv3 kernel traces of the three iteration types -> gpurun_out/trace_TAG.md
tag=${1:-r2}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/trace_$tag.md
echo "# Per-iteration kernel timeline ($tag): rocprofv3 --kernel-trace of tools/mode_trace.py, medians per iteration" > $out
for mode in track geo color; do
  rm -rf /tmp/trace_$mode
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$mode -o t -- python tools/mode_trace.py $mode 40 > /tmp/trace_$mode.log 2>&1
  echo >> $out; grep "host enqueue" /tmp/trace_$mode.log | tail -1 | sed 's/^/`/; s/$/`/' >> $out; echo >> $out
  python tools/trace_summary.py /tmp/trace_$mode "$mode (R = $( [ $mode = track ] && echo 1500 || echo 5000 ), N = 100 000)" >> $out
done
cat $out
