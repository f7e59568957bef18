#!/usr/bin/env python3
"""Round 6: does a hipGraph of one per-frame loop shorten it?  One lk_track_frame call (40 iterations x 4 dependent launches) / one lk_map_frame call captured with
stream capture (torch.cuda.graph on the launch stream: the library enqueues on torch's current stream) and replayed, against the same call enqueued launch by launch.

    python tools/probe/graph_replay.py {track|geo|color} [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loopy_slam_amd import core, workload

mode = sys.argv[1] if len(sys.argv) > 1 else 'track'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = core.Engine()
b = workload.Budget(n_points=100_000)
wl = workload.FrameWorkload(eng, b)
H, W = wl.H, wl.W
e = min(b.ignore_edge, H // 4)
win = (e, H - e, e, W - e)
if mode == 'track':
    rnd = wl._draws(iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]))
    fn = lambda: wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], iters, win, wl.intr, rnd)
else:
    rnd = wl._draws(iters, b.map_rays, H * W)
    fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(eng.device)
    wl.mapper.begin_frame()
    log = eng.zeros(iters, 4)
    wl.mapper.new_frame(wl.rows, None)
    fn = lambda: wl.mapper.run(iters, iters if mode == 'geo' else 0, wl.frames, rnd, fid, (0, H, 0, W), wl.intr, H, W, log)


def timed(f, n=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); f(); e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1) * 1e3 / iters, (t1 - t0) * 1e6 / iters))
    ts.sort()
    return ts[len(ts) // 2]

gpu, host = timed(fn)
print(f'{mode}: launch by launch   {gpu:7.1f} us / iteration on the GPU, host enqueue {host:6.1f} us / iteration')
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    torch.cuda.synchronize()
    gpu_g, host_g = timed(g.replay)
    print(f'{mode}: hipGraph replay     {gpu_g:7.1f} us / iteration on the GPU, host launch  {host_g:6.1f} us / iteration')
except Exception as ex:
    print('capture failed:', repr(ex)[:400])
