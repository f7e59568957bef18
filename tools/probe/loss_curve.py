#!/usr/bin/env python3
"""The losses of ONE mapping call of the benchmark step (24 'geometry' + 36 'color' iterations x 5 000 rays, N = 100 k points, random-init decoders), iteration by iteration -
the round-5 review read `loss_first` 122.8 -> `loss_last` 348.6 as "not a converging optimisation": the first is a 'geometry' loss (depth term), the last a 'color' loss
(depth + 0.1 x colour term over the same rays); inside each stage the loss falls.      python tools/probe/loss_curve.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload

eng = core.Engine()
b = workload.Budget(n_points=100_000)
wl = workload.FrameWorkload(eng, b)
H, W = wl.H, wl.W
for call in range(3):
    rnd = wl._draws(b.map_iters, b.map_rays, H * W)
    fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(eng.device)
    log = eng.zeros(b.map_iters, 4)
    wl.mapper.new_frame(wl.rows, None)
    wl.mapper.run(b.map_iters, b.map_geo_iters, wl.frames, rnd, fid, (0, H, 0, W), wl.intr, H, W, log)
    torch.cuda.synchronize()
    l = log[:, 0].cpu().tolist()
    g, c = l[:b.map_geo_iters], l[b.map_geo_iters:]
    print(f'call {call}: geometry {g[0]:.1f} -> {g[-1]:.1f} (min {min(g):.1f}) | color {c[0]:.1f} -> {c[-1]:.1f} (min {min(c):.1f})')
    print('   ', ' '.join(f'{x:.0f}' for x in l))
