#!/usr/bin/env python3
"""Whole benchmark steps back to back (to be wrapped in `rocprofv3 --kernel-trace`), then tools/probe/step_gaps.py over the trace:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_step -o t -- python tools/probe/step_trace.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
for _ in range(n):
    wl.step()
torch.cuda.synchronize()
