#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for m in geo color track; do
python tools/mode_trace.py $m 40 --repeat 3 2>&1 | tail -1
LK_SERIAL=1 python tools/mode_trace.py $m 40 --repeat 3 2>&1 | tail -1 | sed 's/^/serial: /'
done
