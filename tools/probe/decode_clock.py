#!/usr/bin/env python3
"""Where the cycles of one colour tile of k_decode_fwd go: shader-clock stamps at the phase boundaries of the four waves of the first
workgroups (library built with -DLK_PROBE_CLK: tools/ab_build.sh clk -DLK_PROBE_CLK), tracker-sized (1 500 rays: one tile per
compute unit) and mapper-sized (5 000 rays) launches."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loopy_slam_amd import core, workload

eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
names = ['sample+embed', 'L0 mfma', 'L0 finish', 'barrier', 'L1 mfma', 'L1 finish', 'barrier', 'L2 mfma', 'L2 finish', 'barrier',
         'L3 mfma', 'L3 finish', 'barrier', 'L4 mfma', 'L4 finish', 'output']
def read():
    buf = (C.c_ulonglong * (8 * 4 * 32))()
    assert eng.lib.dll.lk_debug_clk_read(buf) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(8, 4, 32).astype(np.int64)
for label, fn in (('tracker launch (1 500 rays)', lambda: wl.step()), ):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = read()          # the LAST decode_fwd launch of the step = a mapper colour iteration (5 000 rays)
    d = t[:, :, 1:17] - t[:, :, 0:16]
    print('last launch of a step (mapper colour iteration, 5 000 rays): cycles per phase, median over 8 tiles x 4 waves')
    med = np.median(d.reshape(-1, 16), axis=0)
    for n, v in zip(names, med):
        print(f'  {n:14s} {v:9.0f}')
    print('  total', np.median(t[:, :, 16] - t[:, :, 0]))
# tracker-sized: run the tracker alone
b, H, W = wl.b, wl.H, wl.W
e = min(b.ignore_edge, H // 4); win = (e, H - e, e, W - e)
rnd_t = wl._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]))
wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], b.track_iters, win, wl.intr, rnd_t)
torch.cuda.synchronize()
t = read()
d = t[:, :, 1:17] - t[:, :, 0:16]
print('tracker iteration (1 500 rays, one tile per compute unit): cycles per phase, median over 8 tiles x 4 waves')
for n, v in zip(names, np.median(d.reshape(-1, 16), axis=0)):
    print(f'  {n:14s} {v:9.0f}')
print('  total', np.median(t[:, :, 16] - t[:, :, 0]))
