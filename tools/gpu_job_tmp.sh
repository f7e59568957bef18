cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for k in 1 2; do
for v in 0 1 -1; do
LK_SIDE_PRIO=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('prio $v: %.2f ms/step' % d['ms_per_step'], {k[2:]: round(x, 2) for k, x in list(d['kernel_ms_per_step'].items())[:6]})"
done; done
