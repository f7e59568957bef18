#!/usr/bin/env python3
"""Where does the data-parallel code path spend HOST time per mapping iteration?  One rank over RCCL on one GPU (the exchange is then a
copy): lk_map_frame phase 1 / bucket pack / dist.all_reduce / unpack / phase 2, enqueue time of each and wall time per iteration.
    python tools/probe/dist_overhead.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from loopy_slam_amd import core, workload, parallel

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
torch.cuda.set_device(0)
eng = core.Engine()          # before the process group (lk_streams_init)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
T = collections.defaultdict(float)

def timed(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t0; return r
    setattr(obj, name, g)

for mode in ('plain', 'dist'):
    dctx = parallel.DistContext(0, 1) if mode == 'dist' else None
    wl = workload.FrameWorkload(eng, workload.Budget(), dist=dctx)
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    if mode == 'dist':
        timed(eng.lib.dll, 'lk_map_frame', 'lk_map_frame (both phases)')
        timed(eng.lib.dll, 'lk_bucket_copy', 'lk_bucket_copy (pack + unpack)')
        timed(dctx, '_all_reduce', 'dist.all_reduce')
        timed(dctx, 'all_reduce_grads', 'all_reduce_grads (total)')
        timed(wl.mapper, 'run', 'MapOptimizer.run (total)')
        timed(wl.tracker, 'track', 'TrackOptimizer.track (total)')
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{mode}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step')
    if mode == 'dist':
        for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
            print(f'   {k:40s} {1e3 * v / n:7.2f} ms/step host  = {1e6 * v / n / 60:6.1f} us per mapping iteration')
dist.destroy_process_group()
