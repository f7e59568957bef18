#!/usr/bin/env python3
"""The parts of a benchmark step (iterations only), each between two device synchronisations: the tracking call, the row selection +
batch assembly, the mapping call - against the sums of their iterations' steady periods (profiles/r4_iteration_timeline.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload, optim

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
b = wl.b
H, W = wl.H, wl.W
e = min(b.ignore_edge, H // 4)
win = (e, H - e, e, W - e)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
T = {'track': [], 'select+prepare': [], 'map': [], 'whole step (no syncs inside)': []}
for k in range(12):
    rnd_t = wl._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]), wl.gen_track)
    rnd_m = wl._draws(b.map_iters, b.map_rays, H * W)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], b.track_iters, win, wl.intr, rnd_t)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sel = optim.frustum_rows(eng, wl.pos[:wl.n], wl.c2w_host[0], wl.depth_stack[0], wl.intr, H, W, b.frustum_edge, return_mask=True, pending=True)
    prepared = wl.mapper.prepare(b.map_iters, b.map_geo_iters, wl.frames, rnd_m, wl._fid, (0, H, 0, W), wl.intr, H, W, wl.map_log)
    wl.rows, row_mask = sel.finish()
    wl.mapper.new_frame(wl.rows, row_mask, zero=not prepared)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    wl.mapper.run(b.map_iters, b.map_geo_iters, wl.frames, rnd_m, wl._fid, (0, H, 0, W), wl.intr, H, W, wl.map_log)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    T['track'].append(1e3 * (t1 - t0)); T['select+prepare'].append(1e3 * (t2 - t1)); T['map'].append(1e3 * (t3 - t2))
    torch.cuda.synchronize(); t4 = time.perf_counter()
    wl.step()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    T['whole step (no syncs inside)'].append(1e3 * (t5 - t4))
for k, v in T.items():
    v = sorted(v)
    print(f'{k}: median {v[len(v) // 2]:.3f} ms (min {v[0]:.3f})')
