#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for w in 0 1 0 1 0 1; do
export LK_AHEAD2=$w
python tools/probe/step_phases.py 2>/dev/null | tail -2 | tr '\n' ' '; echo " <- LK_AHEAD2=$w"
done
