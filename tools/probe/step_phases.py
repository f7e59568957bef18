#!/usr/bin/env python3
"""Where a benchmark step's wall time goes outside the steady iterations: host-side pieces (draws, frustum selection with its
count read-back, descriptor set-up) against the device time of the two loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload, optim

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
b, H, W = wl.b, wl.H, wl.W
e = min(b.ignore_edge, H // 4)
win = (e, H - e, e, W - e)
sync = torch.cuda.synchronize
acc = {}
def lap(name, t0):
    sync(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
N = 10
for _ in range(N):
    sync(); t = time.perf_counter()
    rnd_t = wl._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2])); t = lap('draws_track', t)
    k = wl.frame_no % b.window
    wl.tracker.track(wl.cam0, wl.depth_stack[k], wl.color_stack[k], b.track_iters, win, wl.intr, rnd_t); t = lap('track', t)
    rnd_m = wl._draws(b.map_iters, b.map_rays, H * W)
    fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(eng.device); t = lap('draws_map', t)
    wl.rows, row_mask = optim.frustum_rows(eng, wl.pos, wl.c2w_stack[k], wl.depth_stack[k], wl.intr, H, W, b.frustum_edge, return_mask=True); t = lap('frustum_rows', t)
    wl.mapper.new_frame(wl.rows, row_mask); t = lap('new_frame', t)
    wl.mapper.run(b.map_iters, b.map_geo_iters, wl.frames, rnd_m, fid, (0, H, 0, W), wl.intr, H, W, wl.map_log); t = lap('map', t)
    wl.frame_no += 1
print({k: round(1e3 * v / N, 3) for k, v in acc.items()}, 'sum %.2f ms' % (1e3 * sum(acc.values()) / N))
sync(); t0 = time.perf_counter()
for _ in range(N):
    wl.step()
sync(); print('step %.2f ms' % (1e3 * (time.perf_counter() - t0) / N))
