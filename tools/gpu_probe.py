#!/usr/bin/env python3
"""Quick GPU probe: forward render at realistic sizes, per-call timing with HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core, profile, synthetic as syn

eng = core.Engine()
print(torch.cuda.get_device_name(0))
W = syn.default_weights()
blob = core.DecoderBlob(eng).pack(W)
NS = [int(x) for x in os.environ.get('PROBE_N', '100000,1000000').split(',')]
RS = os.environ.get('PROBE_R')
for N in NS:
    pos, geo, col = syn.build_cloud(N, device='cpu')
    pos, geo, col = pos.cuda(), geo.cuda(), col.cuda()
    knn = core.KnnIndex(eng, capacity=N)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); knn.build(pos); torch.cuda.synchronize(); t1 = time.perf_counter()
    knn.build(pos); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'N={N}: knn build first {1e3*(t1-t0):.2f} ms, second {1e3*(t2-t1):.3f} ms')
    depth, color, c2w = syn.render_frame(3, device='cuda', holes=0.0)
    H, Wd = depth.shape
    cases = ((1500, True), (5000, True), (5000, False), (10000, False), (H * Wd, True), (H * Wd, False))
    if RS:
        cases = [(int(r), True) for r in RS.split(',')]
    for R, rel in cases:
        cfg = core.RenderCfg(rel_pos=rel)
        g = torch.Generator(device='cpu').manual_seed(R)
        if R == H * Wd:
            jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(Wd, dtype=torch.float32), indexing='ij')
            i, j = ii.reshape(-1).cuda(), jj.reshape(-1).cuda()
        else:
            i = torch.randint(0, Wd, (R,), generator=g).float().cuda(); j = torch.randint(0, H, (R,), generator=g).float().cuda()
        ro, rd = syn.pixel_rays(c2w, i, j)
        gd = depth[j.long(), i.long()].contiguous()
        SAVE = bool(os.environ.get('PROBE_SAVE'))
        st = core.RenderState(eng, R, cfg.S, need_act=SAVE) if SAVE else core.RenderState(eng, R, cfg.S)
        for stage in ('geometry', 'color'):
            for _ in range(3):
                core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, stage, save_act=SAVE)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, stage, save_act=SAVE)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            if os.environ.get('PROBE_KERNELS'):
                kt = profile.KernelTimer(eng, '*'); kt.start()
                for _ in range(n):
                    core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, stage, save_act=SAVE)
                torch.cuda.synchronize()
                print('      ' + '  '.join(f"{k[2:]}={v['total_ms'] / v['calls'] * 1e3:.1f}us" for k, v in kt.stop().items()))
            print(f'  R={R:7d} rel_pos={int(rel)} {stage:8s}: {ms:8.3f} ms  {R/ms/1e3:9.1f} Mrays/s  valid={int(st.valid_ray.sum())} meanhas={float((st.nbr_count>=2).float().mean()):.2f}')
