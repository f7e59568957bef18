#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for w in 100000 512 1024 256 100000 512; do
export LK_SEARCH_WGS=$w
python tools/probe/step_phases.py 2>/dev/null | tail -2 | tr '\n' ' '; echo " <- LK_SEARCH_WGS=$w"
done
