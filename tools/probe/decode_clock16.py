#!/usr/bin/env python3
"""Where the cycles of the 16 x 16 x 32 decoder forward (k_decode_fwd16) go: shader-clock stamps at the phase boundaries of the first
eight colour tiles (eight waves each) and the first 64 geometry waves (library built with -DLK_PROBE_CLK: tools/ab_build.sh clk -DLK_PROBE_CLK).
Printed for the tracker's launch (1 500 rays: one colour tile per compute unit, rel-pos MLP in front) and a mapper colour iteration (5 000 rays)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loopy_slam_amd import core, workload

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
names = ['bias/embed/c + barrier', 'L0 product', 'L0 finish', 'barrier', 'L1 product', 'L1 finish', 'barrier', 'L2 product', 'L2 finish', 'barrier',
         'L3 product', 'L3 finish', 'barrier', 'L4 product', 'L4 finish', 'output']
gnames = ['sample+embed(24 sin)+c', 'L0 product', 'L0 finish+park', 'L1 product', 'L1 finish+park', 'L2 product', 'L2 finish+park',
          'L3 product', 'L3 finish+park', 'L4 product', 'L4 finish']


def read():
    col = (C.c_ulonglong * (8 * 8 * 32))()
    geo = (C.c_ulonglong * (64 * 16))()
    assert eng.lib.dll.lk_debug_clk16_read(col, geo) == 0
    return (np.frombuffer(col, dtype=np.uint64).reshape(8, 8, 32).astype(np.int64), np.frombuffer(geo, dtype=np.uint64).reshape(64, 16).astype(np.int64))


def show(label, relpos):
    t, g = read()
    d = t[:, :, 1:17] - t[:, :, 0:16]
    print(label)
    if relpos:
        print(f'  {"rel-pos MLP phase":24s} {np.median(t[:, :, 18] - t[:, :, 17]):9.0f}   (+ barrier until the decode starts: {np.median(t[:, :, 0] - t[:, :, 18]):.0f})')
    for n, v in zip(names, np.median(d.reshape(-1, 16), axis=0)):
        print(f'  {n:24s} {v:9.0f}')
    print('  colour tile total', np.median(t[:, :, 16] - t[:, :, 0]), ' workgroup entry -> end', np.median(t[:, :, 16] - t[:, :, 17]))
    dg = g[:, 1:12] - g[:, 0:11]
    for n, v in zip(gnames, np.median(dg, axis=0)):
        print(f'  geo {n:24s} {v:9.0f}')
    print('  geometry wave total', np.median(g[:, 11] - g[:, 0]))


for _ in range(2):
    wl.step()
torch.cuda.synchronize()
show('last decode launch of a step = a mapper colour iteration (5 000 rays): cycles per phase, medians over 8 tiles x 8 waves / 64 geometry waves', False)
b, H, W = wl.b, wl.H, wl.W
e = min(b.ignore_edge, H // 4); win = (e, H - e, e, W - e)
rnd_t = wl._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]))
wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], b.track_iters, win, wl.intr, rnd_t)
torch.cuda.synchronize()
show('tracker iteration (1 500 rays, one colour tile per compute unit)', True)
