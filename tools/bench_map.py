#!/usr/bin/env python3
"""Timing of the map-maintenance kernels on one MI355X: frustum row selection over N points and
point insertion of n rays against a cloud of N points.  Wall time of the whole call including its one host sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core, optim, synthetic as syn

eng = core.Engine()
I = syn.TUM_INTR
intr = (I['fx'], I['fy'], I['cx'], I['cy'])
depth, _, c2w = syn.render_frame(3, device='cuda', holes=0.02)
print('| kernel | N points | rays | selected / accepted | ms | algorithmic GB/s |')
print('|---|---|---|---|---|---|')
for N in (100_000, 1_000_000, 5_000_000):
    pos, _, _ = syn.build_cloud(N, device='cpu')
    pos = pos.cuda()
    for _ in range(2):
        rows = optim.frustum_rows(eng, pos, c2w, depth, intr, I['H'], I['W'], -4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        rows = optim.frustum_rows(eng, pos, c2w, depth, intr, I['H'], I['W'], -4)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    # per point: 2 x 12 B position reads (two passes) + 16 B depth taps + 4 B sampled depth written and read + 1 B mask
    # written and read + 4 B index written for selected rows
    gb = (N * (24 + 16 + 8 + 2) + rows.numel() * 4) / 1e9
    print(f'| lk_frustum_rows | {N:,} |  | {rows.numel():,} | {ms:.3f} | {gb / ms * 1e3:.0f} |')
    knn = core.KnnIndex(eng, capacity=N)
    knn.build(pos)
    n = 6000 if N < 1_000_000 else 60000
    g = torch.Generator().manual_seed(1)
    c2 = syn.loop_pose(37, 200, 'cpu')
    ro, rd = syn.pixel_rays(c2, torch.rand(n, generator=g) * (I['W'] - 1), torch.rand(n, generator=g) * (I['H'] - 1))
    gd = syn.room_depth(ro, rd)
    ro, rd, gd = ro.cuda(), rd.cuda(), gd.cuda()
    for _ in range(2):
        acc, pts = optim.add_points(eng, knn, ro, rd, gd, 0.04 ** 2, 0.98, 1.02)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        acc, pts = optim.add_points(eng, knn, ro, rd, gd, 0.04 ** 2, 0.98, 1.02)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f'| lk_add_points | {N:,} | {n:,} | {acc.numel():,} | {ms:.3f} |  |')
