#!/usr/bin/env python3
"""SURVEY §8(d) measurement grid on one MI355X: modes x N points x R rays, synthetic 640x480 room.

    python tools/sweep.py [--points 100000,500000,2000000] [--md profiles/r1_sweep.md]

Modes: render (forward only, full image = Renderer.render_img), map-geometry / map-color (one joint mapping iteration of a
24-iteration lk_map_frame call: batch assembly, search, forward, loss, backward, Adam), track (one tracking iteration).  Everything resident in HBM, HIP-event
timing of `iters` back-to-back iterations after a warm-up.  'HBM frac' = 11.1 KB/ray (forward
algorithmic bytes) x rays/s / 8 TB/s, quoted for the forward-only mode as the north star asks; the forward is
compute-bound (100-180 FLOP/B), so the fp32 matrix fraction (forward FLOPs/ray x rays/s / 157.3 TFLOP/s) is beside it.
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core, workload, synthetic as syn


def timed(fn, iters, warm=2):
    """Median over `iters` calls, each between its own pair of events (one call that lands behind a free / a first-use allocation of the process -
    the r6 grid first had ONE cell at 15 x its neighbours - does not set the figure)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', default='100000,500000,2000000,5000000')
    ap.add_argument('--dense', default='2000000', help='sizes that are ALSO measured as over-dense stress clouds (all points in one room)')
    ap.add_argument('--md', default=None)
    args = ap.parse_args()
    eng = core.Engine()
    rows = []
    sizes = [(int(x), True) for x in args.points.split(',')] + [(int(x), False) for x in args.dense.split(',') if x]
    for N, online in sizes:
        for rel in (True, False):
            cfgname = ('Replica (rel-pos MLP)' if rel else 'TUM/ScanNet (plain colour)') + ('' if online else ' - STRESS: one room')
            for (tr, mr) in ((1500, 5000), (5000, 10000)):
                if rel != (tr == 1500):
                    continue                      # Replica budget with the rel-pos model, TUM/ScanNet budget without
                b = workload.Budget(n_points=N, track_rays=tr, map_rays=mr, rel_pos=rel, online_cloud=online)
                wl = workload.FrameWorkload(eng, b)
                H, W = wl.H, wl.W
                e = min(b.ignore_edge, H // 4)
                win = (e, H - e, e, W - e)
                fid = (torch.arange(mr, dtype=torch.int32) % b.window).to(eng.device)
                n_it = 24
                rnd_m = wl._draws(n_it, mr, H * W)
                log = eng.zeros(n_it, 4)
                # the mapping loop as the product runs it: ONE lk_map_frame call of n_it iterations of one stage
                for stage, n_geo in (('geometry', n_it), ('color', 0)):
                    ms = timed(lambda: wl.mapper.run(n_it, n_geo, wl.frames, rnd_m, fid, (0, H, 0, W), wl.intr, H, W, log), 3, warm=1) / n_it
                    rows.append((cfgname, wl.n, f'map-{stage}', mr, ms, mr / ms * 1e3, None))
                rnd_t = wl._draws(10, tr, (win[1] - win[0]) * (win[3] - win[2]))
                ms = timed(lambda: wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], 10, win, wl.intr, rnd_t), 2, warm=1) / 10
                rows.append((cfgname, wl.n, 'track', tr, ms, tr / ms * 1e3, None))
                # forward only, full image
                R = H * W
                jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
                ro, rd = syn.pixel_rays(wl.c2w_stack[0], ii.reshape(-1).to(eng.device), jj.reshape(-1).to(eng.device))
                gd = wl.depth_stack[0].reshape(-1).contiguous()
                st = core.RenderState(eng, R, wl.cfg.S)
                ms = timed(lambda: core.render_forward(eng, wl.cfg, st, ro, rd, gd, wl.knn, wl.pos, wl.geo, wl.col, wl.dec, 'color'), 5)
                flop_ray = 1.99e6 if rel else 1.13e6          # BASELINE.md §3, forward FLOPs per ray
                rows.append((cfgname, wl.n, 'render (fwd)', R, ms, R / ms * 1e3, (11.1e3 * R / (ms * 1e-3) / 8e12, flop_ray * R / (ms * 1e-3) / 157.3e12)))
                del wl, st
                torch.cuda.empty_cache()
    lines = ['| config | N points | mode | rays / iteration | ms / iteration | M rays/s | HBM frac (11.1 KB/ray) | fp32-MFMA frac (1.99 / 1.13 MFLOP/ray) |', '|---|---|---|---|---|---|---|---|']
    for c, N, mode, R, ms, rps, frac in rows:
        lines.append(f'| {c} | {N:,} | {mode} | {R:,} | {ms:.3f} | {rps / 1e6:.2f} | {"" if frac is None else f"{frac[0]:.3f}"} | {"" if frac is None else f"{frac[1]:.3f}"} |')
    out = '\n'.join(lines)
    print(out)
    if args.md:
        with open(args.md, 'w') as f:
            f.write('# Measurement grid (SURVEY §8d), one MI355X, synthetic 640x480 rooms, fp32 (map modes: per iteration of a 24-iteration lk_map_frame call)\n\n'
                    '`python tools/sweep.py` - HIP-event timing, everything resident in HBM.  Maps are laid down the way an online run does it\n'
                    '(synthetic.build_cloud_online: radius-de-duplicated insertion with lk_add_points, ~900 points per m^2 of surface) and grow\n'
                    'with the EXPLORED AREA: one 6 x 4 x 3 m room per 100 000 points (the camera works in the last room), as a long sequence grows\n'
                    'its map.  Rows marked STRESS pack all points into one room instead (random pixels, no de-duplication: 20x the density an\n'
                    'online map can reach at radius_add 0.04) - a stress axis for the radius search, not a map the system produces.\n\n' + out + '\n')


if __name__ == '__main__':
    main()
