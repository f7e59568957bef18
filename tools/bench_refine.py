#!/usr/bin/env python3
"""Whole-map refinement iterations (BASELINE config 5: every row of the map trainable, colour decoder frozen, Mapper.py:884-897) at N points:
ms per 'geometry' and per 'color' iteration of lk_map_frame with rows = NULL.

    python tools/bench_refine.py [--points 5000000] [--rays 10000] [--iters 40] [--rel-pos]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core, steps, workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=5_000_000)
    ap.add_argument('--rays', type=int, default=10_000)
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--rel-pos', action='store_true')
    ap.add_argument('--f16', action='store_true', help='half feature tables (config 5)')
    a = ap.parse_args()
    eng = core.Engine()
    b = workload.Budget(n_points=a.points, rel_pos=a.rel_pos)
    b.map_rays = a.rays
    wl = workload.FrameWorkload(eng, b)
    geo, col = (wl.geo.half(), wl.col.half()) if a.f16 else (wl.geo, wl.col)
    H, W = wl.H, wl.W
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    mo = steps.MapOptimizer(eng, wl.cfg, wl.dec, wl.knn, wl.pos, geo, col, None, a.rays, lrs, w_color=0.1, fix_color_decoder=True)
    fid = (torch.arange(a.rays, dtype=torch.int32) % b.window).to(eng.device)
    rnd = wl._draws(a.iters, a.rays, H * W)
    log = eng.zeros(a.iters, 4)
    out = {}
    for stage, n_geo in (('geometry', a.iters), ('color', 0)):
        for rep in range(3):
            mo.new_frame(None, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mo.run(a.iters, n_geo, wl.frames, rnd, fid, (0, H, 0, W), wl.intr, H, W, log)
            torch.cuda.synchronize()
            out[stage] = 1e3 * (time.perf_counter() - t0) / a.iters
    print(f'whole-map refinement, N = {wl.n} points ({"fp16" if a.f16 else "fp32"} tables), {a.rays} rays, '
          f'{"rel-pos" if a.rel_pos else "plain"} colour model: geometry {out["geometry"]:.3f} ms / iteration, color {out["color"]:.3f} ms / iteration')


if __name__ == '__main__':
    main()
