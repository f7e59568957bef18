#!/usr/bin/env python3
"""Upper bound for half-batch pipelining: two independent mapper loops of R/2 rays on two streams of one process (two host
threads, LK_SERIAL=1 so that the library's own side streams stay out of it) against one loop of R rays."""
import os, sys, threading, time
os.environ['LK_SERIAL'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload

def make(rays):
    eng = core.Engine()
    b = workload.Budget()
    b.map_rays = rays
    wl = workload.FrameWorkload(eng, b)
    H, W = wl.H, wl.W
    iters = 40
    rnd = wl._draws(iters, b.map_rays, H * W)
    fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(eng.device)
    wl.mapper.begin_frame()
    log = eng.zeros(iters, 4)
    def fn():
        wl.mapper.new_frame(wl.rows, None)
        wl.mapper.run(iters, 0, wl.frames, rnd, fid, (0, H, 0, W), wl.intr, H, W, log)
    return fn, iters

def timed(fns, streams, reps=4):
    def work(fn, st):
        with torch.cuda.stream(st):
            for _ in range(reps):
                fn()
    for fn, st in zip(fns, streams):
        work(fn, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(fn, st)) for fn, st in zip(fns, streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

f1, it = make(5000)
s0 = torch.cuda.Stream()
print('one loop of 5000 rays (serial mode): %.1f us/iteration' % (1e6 * timed([f1], [s0]) / it))
fa, _ = make(2500); fb, _ = make(2500)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print('one loop of 2500 rays: %.1f us/iteration' % (1e6 * timed([fa], [sa]) / it))
print('two loops of 2500 rays on two streams: %.1f us/iteration pair' % (1e6 * timed([fa, fb], [sa, sb]) / it))
