#!/usr/bin/env python3
"""Diagnosis: tests/test_parity_at_size.py::test_tracker_iteration_vs_oracle_at_bench_size on a box where it fails - which rays carry the
gradient difference against the float64 referee, and what is special about them (neighbours near the radius edge, loss-mask threshold,
ReLU gates)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import atsize as A
from oracle import hotpath as H
from loopy_slam_amd import _ffi, core, optim, synthetic as syn

torch.set_num_threads(16)
model, R = (sys.argv[1] if len(sys.argv) > 1 else 'replica'), int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rel = model == 'replica'
eng = core.Engine()
pos, geo, col = A.scene(100_000)
W = syn.default_weights(rel_pos=rel)
dpos, dgeo, dcol = eng.f32(pos), eng.f32(geo), eng.f32(col)
knn = core.KnnIndex(eng, capacity=100_000); knn.build(dpos)
dec = core.DecoderBlob(eng).pack(W)
b = A.ray_batch(R, frame=5, holes=0.0, seed=2, window=(100, A.I['H'] - 100, 100, A.I['W'] - 100))
cam = H.c2w_to_cam(b['c2w'])
cfg = core.RenderCfg(rel_pos=rel)
st = core.RenderState(eng, R, cfg.S, need_act=True)
dcam, pi, pj = eng.f32(cam), eng.f32(b['i']), eng.f32(b['j'])
ro, rd = eng.empty(R, 3), eng.empty(R, 3)
optim.rays_from_pose(eng, dcam, pi, pj, A.INTR, ro, rd)
gd, gc = eng.f32(b['gt_depth']), eng.f32(b['gt_color'])
core.render_forward(eng, cfg, st, ro, rd, gd, knn, dpos, dgeo, dcol, dec, 'color', tracker=True, save_act=True, extra_flags=_ffi.FLAG_ZERO_ABSENT)
d_depth, d_color, out4 = eng.empty(R), eng.empty(R, 3), eng.zeros(4)
optim.loss_tracker(eng, st, gd, gc, 0.5, True, d_depth, d_color, out4, eng.empty(R + 8))
torch.cuda.synchronize()
bo = dict(b); bo['rays_o'], bo['rays_d'] = ro.cpu(), rd.cpu()
z, _ = H.sample_z(b['gt_depth'], 0.98, 1.02, 0.3, 5)
p = H.sample_points(bo['rays_o'], bo['rays_d'], z)
got = st.nbr_idx.cpu().numpy()
d2, idx, cnt, n_re = A.contract_knn(pos, p, np.float32(0.08 ** 2), got_idx=got)
print('knn equal', np.array_equal(got, idx), 'rechecked', n_re)
kn = (d2, idx, cnt)
ro_o, rd_o = H.rays_from_uv(b['i'], b['j'], H.quat_to_c2w(cam), *A.INTR)
with torch.no_grad():
    o0 = H.render_batch(A.ocfg(rel), ro_o, rd_o, b['gt_depth'], pos, geo, col, W, 'color', tracker=True, knn=kn)
bp, margin = A.branch_point_rays(o0, b, pos, geo, W, tracker_loss=True)
print('branch point rays', int(bp.sum()), 'relu margin', margin)
if int(bp.sum()):
    d_depth[bp.to(eng.device)] = 0.0; d_color[bp.to(eng.device)] = 0.0
gs = core.GradState(eng, pos.shape[0], R, dec.n, feats=False, weights=False, rays=True)
core.render_backward(eng, st, gs, d_depth, d_color)
torch.cuda.synchronize()
kr = (ro.cpu(), rd.cpu())
r = A.oracle_tracker(rel, b, cam, pos, geo, col, W, kn, exclude=bp, rays_value=kr)
r64 = A.oracle_tracker64(rel, b, cam, pos, geo, col, W, kn, exclude=bp, var32=r['out']['var'].detach(), rays_value=kr)
for name, got_g, o32, f64 in (('rays_o', gs.g_rays_o.cpu(), r['g_rays_o'], r64['g_rays_o']), ('rays_d', gs.g_rays_d.cpu(), r['g_rays_d'], r64['g_rays_d'])):
    s = float(f64.abs().max())
    eh, eo = (got_g.double() - f64).abs().max(1).values / s, (o32.double() - f64).abs().max(1).values / s
    top = torch.argsort(eh, descending=True)[:6]
    print(name, 'scale', s, 'hip max %.2e o32 max %.2e' % (float(eh.max()), float(eo.max())), 'rays with err > 1e-5:', int((eh > 1e-5).sum()))
    for t in top.tolist():
        samp = slice(5 * t, 5 * t + 5)
        d2r = torch.from_numpy(d2[samp]); r2 = np.float32(0.08 ** 2)
        edge = float(((d2r - r2).abs()[torch.from_numpy(idx[samp]) >= 0]).min() / r2) if (idx[samp] >= 0).any() else -1
        tmp = float(abs(b['gt_depth'][t] - o0['depth'][t]) / torch.sqrt(o0['var'][t] + 1e-10))
        print('   ray %d err hip %.2e o32 %.2e | counts %s | min |d2-r2|/r2 %.2e | var %.3e tmp %.3f | |g| %.2e | hip-vs-o32 depth %.2e var rel %.2e' %
              (t, float(eh[t]), float(eo[t]), cnt[samp].tolist(), edge, float(o0['var'][t]), tmp, float(f64[t].abs().max()) / s,
               float(abs(st.depth.cpu()[t] - o0['depth'][t])), float(abs(st.var.cpu()[t] - o0['var'][t]) / (o0['var'][t] + 1e-30))))
thr = 10 * (torch.abs(b['gt_depth'] - o0['depth']) / torch.sqrt(o0['var'] + 1e-10)).mean()
print('loss mask threshold', float(thr), 'loss hip', float(out4[0]), 'masked', int(out4[3]))
import platform, subprocess
print(subprocess.run("lscpu | grep -E 'Model name|Flags' | cut -c1-300", shell=True, capture_output=True, text=True).stdout[:600])
