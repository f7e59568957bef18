#!/bin/bash
# tools/collect_round.sh TAG : after tools/gpu_job_<TAG>_final.sh has come back, copy what the judged summaries are made of from gpurun_out/ (scratch)
# into profiles/ (tracked).  tools/summarize_prof.py TAG writes profiles/TAG_rocprof_summary.md + TAG_kernel_stats.csv itself.
tag=${1:-r5}
cd "$(dirname "$0")/.."
python tools/summarize_prof.py $tag > /dev/null
cp gpurun_out/bench_$tag.json profiles/${tag}_bench.json
cp gpurun_out/trace_$tag.md profiles/${tag}_iteration_timeline.md
cp gpurun_out/sq_$tag.txt profiles/${tag}_sq_counters.md
{ echo "# One iteration of each type launch by launch ($tag closing run): rocprofv3 --kernel-trace of tools/mode_trace.py, the iteration of median length;"
  echo "# us from the iteration's first launch; queue = HIP stream (queue 1 = the weight-gradient stream: in a 'color' iteration it also runs the colour trunk's"
  echo "# half of the step - k_bwd_reduce (tile sums + fc_c products + Adam rider) and k_repack_trunk - joined in front of the NEXT iteration's k_decode_fwd)"
  for m in track geo color; do echo; echo "## $m"; echo; cat gpurun_out/gantt_${tag}_$m.md; done; } > profiles/${tag}_gantt.md
for n in room scene0000 freiburg1_desk; do cp gpurun_out/slam_run_$n.json profiles/${tag}_slam_run_$n.json; done
for f in loops_at_size teacher_forced bench_dist_one_rank; do      # (compact: the indented teacher-forced record is 6 000 lines)
  [ -f gpurun_out/$f.json ] && python -c "import json; json.dump(json.load(open('gpurun_out/$f.json')), open('profiles/${tag}_$f.json', 'w'), separators=(',', ':'))"
done
ls -la profiles/${tag}_* | awk '{print $5, $9}'
# round 6: per-stage counter traffic (the roofline's measured bytes), single-stream colour iteration, tracking chain, measurement grid, accuracy
[ -f gpurun_out/stage_traffic_$tag.json ] && cp gpurun_out/stage_traffic_$tag.json profiles/${tag}_stage_traffic.json && cp gpurun_out/stage_traffic_$tag.md profiles/${tag}_color_iteration_traffic.md
[ -f gpurun_out/gantt_${tag}_color_serial.md ] && cp gpurun_out/gantt_${tag}_color_serial.md profiles/${tag}_color_iteration_one_stream.md
[ -f gpurun_out/track_chain_$tag.md ] && cp gpurun_out/track_chain_$tag.md profiles/${tag}_track_chain.md
[ -f gpurun_out/sweep_$tag.md ] && cp gpurun_out/sweep_$tag.md profiles/${tag}_sweep.md
[ -f gpurun_out/gpu_tests_$tag.log ] && tail -n 12 gpurun_out/gpu_tests_$tag.log > profiles/${tag}_gpu_tests.txt
ls gpurun_out/accuracy_*.json > /dev/null 2>&1 && python tools/accuracy_summary.py $tag > /dev/null
ls -la profiles/${tag}_* | awk '{print $5, $9}'
