#!/usr/bin/env python3
"""The full step's two modes (18.3 / 19.0 ms per step between runs of the same library on the same box): per-step durations from events on the
launch stream at every step boundary, and the full-frame render's own span on its stream - which steps carry the difference.
    python tools/probe/full_step_modes.py [steps (default 20)]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
wl.step(); wl.step()
wl.frame_no = 0
wl.step(full=True)
wl.frame_no = 0
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
rs = wl.render_stream
rev = []
orig = wl.render_frame


def timed_render(k):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream()); r = orig(k); b.record(torch.cuda.current_stream())
    rev.append((a, b))
    return r


wl.render_frame = timed_render
t0 = time.perf_counter()
ev[0].record()
host = []
for i in range(n):
    h0 = time.perf_counter()
    wl.step(full=True)
    host.append(1e3 * (time.perf_counter() - h0))
    ev[i + 1].record()
torch.cuda.synchronize()
dt = 1e3 * (time.perf_counter() - t0) / n
steps = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print('ms/step %.2f  | per step (device, launch stream):' % dt, ' '.join('%.1f' % s for s in steps))
print('   host enqueue per step:', ' '.join('%.1f' % s for s in host))
print('   render spans on their stream (ms):', ' '.join('%.2f' % a.elapsed_time(b) for a, b in rev),
      ' starts after step-0 start (ms):', ' '.join('%.1f' % ev[0].elapsed_time(a) for a, b in rev))
