#!/bin/bash
# tools/ab_env.sh N VAR  : N alternating bench runs with and without the environment switch VAR=1 (same library, same box)
cd "$(dirname "$0")/.."
n=$1; var=$2
for k in $(seq $n); do for v in 0 1; do
  if [ $v = 1 ]; then export $var=1; else unset $var; fi
  python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; print('$var=$v %.2f ms/step  decode_bwd %.3f' % (d['ms_per_step'], k.get('k_decode_bwd', 0)))"
done; done
