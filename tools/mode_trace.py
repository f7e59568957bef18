#!/usr/bin/env python3
"""Run ONE iteration type of the benchmark workload back to back (to be wrapped in `rocprofv3 --kernel-trace`):

    python tools/mode_trace.py {track|geo|color} [iterations] [--points N] [--graph]

tools/trace_summary.py turns the dispatch trace into per-kernel durations and the gaps between kernels of one iteration."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core, workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('mode', choices=('track', 'geo', 'color'))
    ap.add_argument('iters', type=int, nargs='?', default=40)
    ap.add_argument('--points', type=int, default=100_000)
    ap.add_argument('--repeat', type=int, default=3)
    ap.add_argument('--rays', type=int, default=0, help='override the mapping batch size')
    ap.add_argument('--geo-free', action='store_true', help='mapping.fix_geo_decoder: False - the geometry decoder is trained too (k_geo_wgrad)')
    args = ap.parse_args()
    eng = core.Engine()
    b = workload.Budget(n_points=args.points)
    if args.rays:
        b.map_rays = args.rays
    wl = workload.FrameWorkload(eng, b)
    if args.geo_free:
        from loopy_slam_amd import steps
        wl.mapper = steps.MapOptimizer(eng, wl.cfg, wl.dec, wl.knn, wl.pos, wl.geo, wl.col, wl.rows, b.map_rays, workload.MAP_LRS, w_color=0.1, fix_geo_decoder=False)
    H, W = wl.H, wl.W
    e = min(b.ignore_edge, H // 4)
    win = (e, H - e, e, W - e)
    if args.mode == 'track':
        rnd = wl._draws(args.iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]))
        fn = lambda: wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], args.iters, win, wl.intr, rnd)
    else:
        stage = 'geometry' if args.mode == 'geo' else 'color'
        rnd = wl._draws(args.iters, b.map_rays, H * W)
        fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(eng.device)
        wl.mapper.begin_frame()

        log = eng.zeros(args.iters, 4)

        def fn():
            wl.mapper.new_frame(wl.rows, None)
            wl.mapper.run(args.iters, args.iters if stage == 'geometry' else 0, wl.frames, rnd, fid, (0, H, 0, W), wl.intr, H, W, log)
    fn()
    torch.cuda.synchronize()
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'{args.mode}: host enqueue {1e6 * (t1 - t0) / args.iters:.1f} us/iteration, wall {1e6 * (t2 - t0) / args.iters:.1f} us/iteration', flush=True)


if __name__ == '__main__':
    main()
