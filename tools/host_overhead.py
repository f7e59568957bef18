#!/usr/bin/env python3
"""How long does the HOST take to enqueue one benchmark step (no sync) vs the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core, workload

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
wl.step(); torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms')
# tracker only / mapper only
b = wl.b
import types
for name, fn in (('track', lambda: wl._track()), ('map', lambda: wl._map())):
    if not hasattr(wl, '_' + name):
        break
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{name}: enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms')
