#!/bin/bash
# tools/ab_envval.sh N VAR VAL [VAL...] : N alternating headline bench runs per value of the environment variable VAR ("-" = unset), same library, same box
cd "$(dirname "$0")/.."
n=$1; var=$2; shift; shift
for k in $(seq $n); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; print('$var=$v %.3f ms/step  iterations %.3f  wgrad %.3f gather %.3f relpos_bwd %.3f' % (d['ms_per_step'], d['ms_per_step_iterations'], k.get('k_wgrad', 0), k.get('k_feat_gather', 0), k.get('k_relpos_bwd', 0)))"
done; done
