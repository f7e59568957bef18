#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { timeout 600 python tools/accuracy_run.py --pipeline hip "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k in ('ate_rmse_cm', 'rot_err_deg', 'depth_l1_cm', 'wall_s', 'points', 'max_translation_error_cm')})"; }
for s in 1219 1220 1221; do echo "== TUM 2000 rays iters x0.5 seed $s"; run --config configs/TUM_RGBD/freiburg1_desk.yaml --frames 50 --rays 2000 --iters-scale 0.5 --color-refine 0 --seed $s; done
for s in 1219 1220 1221; do echo "== ScanNet 2000 rays seed $s"; run --config configs/ScanNet/scene0000.yaml --frames 50 --rays 2000 --color-refine 0 --seed $s; done
for s in 1219 1220; do echo "== ScanNet 3000 rays seed $s"; run --config configs/ScanNet/scene0000.yaml --frames 50 --rays 3000 --color-refine 0 --seed $s; done
