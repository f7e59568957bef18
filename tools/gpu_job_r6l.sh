#!/bin/bash
# round-6 job L: can k_wgrad and k_relpos_bwd_fused be made CO-RESIDENT (each at one wave per SIMD: 256 + 216 registers) instead of queueing behind one
# another (two relpos workgroups per unit = the whole register file)?  x1a: 256 relpos workgroups + 1 024 wgrad waves, x1b: 256 + 2 048, x1c: 512 + 1 024
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/ab_full.sh 2 head x1a x1b x1c
