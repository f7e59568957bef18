#!/usr/bin/env python3
"""What a mapped frame of the benchmark's full step costs beside its iterations: insertion + feature rows + index rebuild, and the full-frame
render, each timed alone (device idle before and after)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
for _ in range(3):
    wl.step(full=True)
torch.cuda.synchronize()
ta, tr = [], []
for k in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wl.mapped_frame_extras(k % wl.b.window)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    wl.render_frame(k % wl.b.window)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ta.append(1e3 * (t1 - t0)); tr.append(1e3 * (t2 - t1))
print('insertion + rows + index rebuild: ms', [round(x, 3) for x in ta])
print('full-frame render (307 200 rays): ms', [round(x, 3) for x in tr])
