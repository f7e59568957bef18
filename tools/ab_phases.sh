#!/bin/bash
# tools/ab_phases.sh NAME...  (on the GPU box): step-phase times (tools/probe/step_phases.py) for each ab/lib_NAME.so
cd "$(dirname "$0")/.."
for v in "$@"; do
  cp ab/lib_$v.so loopy_slam_amd/libloopyhip.so
  python tools/probe/step_phases.py 2>/dev/null | tail -2 | tr '\n' ' '; echo " <- $v"
done
