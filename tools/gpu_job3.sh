#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
run() {
  LK_SERIAL=1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; print('A=$LK_WG_COST_A B=$LK_WG_COST_B C=$LK_WG_COST_C serial wgrad %.3f  step %.2f' % (k['k_wgrad'], d['ms_per_step']))"
}
ov() {
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; print('A=$LK_WG_COST_A B=$LK_WG_COST_B C=$LK_WG_COST_C overlapped step %.2f wgrad %.2f' % (d['ms_per_step'], k['k_wgrad']))"
}
export LK_WG_COST_A=0 LK_WG_COST_B=1 LK_WG_COST_C=0; run; ov
export LK_WG_COST_A=0 LK_WG_COST_B=1 LK_WG_COST_C=-0.5; run
export LK_WG_COST_A=0 LK_WG_COST_B=1 LK_WG_COST_C=-1; run
export LK_WG_COST_A=0 LK_WG_COST_B=1 LK_WG_COST_C=0.5; run
export LK_WG_COST_A=-0.25 LK_WG_COST_B=1 LK_WG_COST_C=0; run
export LK_WG_COST_A=0.25 LK_WG_COST_B=1 LK_WG_COST_C=0; run; ov
export LK_WG_COST_A=1 LK_WG_COST_B=0 LK_WG_COST_C=0.5; ov
export LK_WG_COST_A=0 LK_WG_COST_B=1 LK_WG_COST_C=0; ov
