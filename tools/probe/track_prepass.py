#!/usr/bin/env python3
"""What a tracked frame of the TUM / ScanNet configs costs in front of its iterations (the per-frame image pre-pass and the per-iteration
pixel draws from the gradient pool, Tracker.py:126-139, 243-268), each step timed alone on the device."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, optim, synthetic as syn

eng = core.Engine()
H, W = 480, 640
depth, color, _ = syn.render_frame(3, motion='handheld', scene='furnished')
color, depth = color.to(eng.device).float().contiguous(), depth.to(eng.device).float().contiguous()
n_px, iters = 5000, 200
win = (20, H - 20, 20, W - 20)
gen = torch.Generator(device=eng.device).manual_seed(1)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, out


t1, (grad, _, r2q) = timed(lambda: optim.radius_maps(eng, color, 0.02, 0.08, 0.02, 2.0))
t2, pool = timed(lambda: optim.top_grad_pixels(eng, grad, 15 * n_px, win, depth, False))
t3, u = timed(lambda: torch.rand(iters, pool.numel(), generator=gen, device=eng.device))
t4, order = timed(lambda: u.topk(n_px, dim=1).indices)
t5, rnd = timed(lambda: pool[order].contiguous())
print(f'radius maps {t1:.3f} ms | top-gradient pool ({pool.numel()} pixels) {t2:.3f} ms | uniform draws [{iters} x {pool.numel()}] {t3:.3f} ms | '
      f'topk {n_px} per row {t4:.3f} ms | gather {t5:.3f} ms')
