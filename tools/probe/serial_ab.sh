cd /root/repo
for k in 1 2 3 4; do
  for m in overlap serial; do
    if [ $m = serial ]; then export LK_SERIAL=1; else unset LK_SERIAL; fi
    python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('$m %.2f  %s %.1fus frac %.3f' % (d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac']))"
  done
done
