#!/usr/bin/env python3
"""Is the split step (LkBwdExtra::split_reduce) race-free?  The at-size mapping call of tests/test_loops_at_size.py, repeated K times per
library variant in ONE process each (ab/lib_split.so = the shipped library, ab/lib_nosplit.so = the same sources with `ex.split_reduce = 0` in
lk_loop.hip - round 5 built it with a -DLK_SPLIT_STEP=0 macro that has since been removed), the per-iteration losses against the oracle loop's
(computed once).  A stale read behind the moved join would show as an occasional jump of the loss difference right after the first 'color'
iterations; chaos alone grows smoothly from ~1e-5 there.

    python tools/probe/split_step_race.py K          (on the GPU box; prints one line per run)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def child(k_runs, oracle_file):
    import numpy as np
    import torch
    import atsize as A
    import test_loops_at_size as L
    from oracle import hotpath as H
    from loopy_slam_amd import core, steps, synthetic as syn
    torch.set_num_threads(16)
    N, R, iters, n_geo, window = 100_000, 5000, 60, 24, 12
    pos, geo, col = A.scene(N)
    W = syn.default_weights(rel_pos=True)
    fr = [syn.render_frame(3 * k, device='cpu', holes=0.02) for k in range(window)]
    depth_s, color_s, pose_s = (torch.stack([f[q] for f in fr]).contiguous() for q in range(3))
    Hh, Ww = depth_s.shape[1:]
    g = torch.Generator().manual_seed(5000 + R)
    rnd_all = torch.randint(0, Hh * Ww, (iters, R), generator=g, dtype=torch.int32)
    fid = (torch.arange(R) % window).long()
    rows = torch.from_numpy(H.frustum_rows(pos.numpy(), pose_s[0].numpy(), depth_s[0].numpy(), *A.INTR, Hh, Ww, -4)).long()
    if not os.path.exists(oracle_file):
        dec_names = list(steps.GEO_DECODER_PARAMS) + [n for n in steps.COLOR_DECODER_PARAMS if n in W]
        ol = L.oracle_map_loop(L.TreeRender(pos, True), geo, col, W, rows, (depth_s, color_s, pose_s), fid, rnd_all, n_geo, A.INTR, L.MAP_LRS, dec_names)[0]
        np.save(oracle_file, np.array(ol))
    ol = np.load(oracle_file)
    eng = core.Engine()
    dpos = eng.f32(pos)
    knn = core.KnnIndex(eng, capacity=N)
    knn.build(dpos)
    mask = torch.zeros(N, dtype=torch.uint8)
    mask[rows] = 1
    frames = (eng.f32(depth_s), eng.f32(color_s), eng.f32(pose_s), None)
    first = None
    for k in range(k_runs):
        dec = core.DecoderBlob(eng).pack(W)
        dgeo, dcol = eng.f32(geo).clone(), eng.f32(col).clone()
        mo = steps.MapOptimizer(eng, core.RenderCfg(rel_pos=True), dec, knn, dpos, dgeo, dcol, None, R, L.MAP_LRS, w_color=0.1)
        mo.new_frame(rows.to(torch.int32).to(eng.device), mask.to(eng.device))
        log = eng.zeros(iters, 4)
        mo.run(iters, n_geo, frames, rnd_all.to(eng.device), fid.to(torch.int32).to(eng.device), (0, Hh, 0, Ww), A.INTR, Hh, Ww, log)
        torch.cuda.synchronize()
        kl = log[:, 0].cpu().numpy().astype(np.float64)
        rel = np.abs(kl - ol) / np.abs(ol)
        first = kl if first is None else first
        rr = np.abs(kl - first) / np.abs(first)
        print('  run %d  vs oracle it24..31: %s  max %.1e | vs run 0: it24..31 max %.1e, all %.1e' %
              (k, ' '.join('%.1e' % x for x in rel[24:32]), rel.max(), rr[24:32].max(), rr.max()), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), sys.argv[3])
    else:
        k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
        of = '/tmp/split_race_oracle.npy'
        for rep in range(2):
            for v in ('split', 'nosplit'):
                subprocess.check_call(['cp', os.path.join(ROOT, 'ab', f'lib_{v}.so'), os.path.join(ROOT, 'loopy_slam_amd', 'libloopyhip.so')])
                print(f'== {v}', flush=True)
                subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', str(k), of])
