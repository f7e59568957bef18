import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import atsize as A
from oracle import hotpath as H
from loopy_slam_amd import _ffi, core, synthetic as syn
from util import make_engine
import test_parity_at_size as T
torch.set_num_threads(16)
print('cpu flags avx512:', 'avx512f' in open('/proc/cpuinfo').read(), torch.__config__.show().split('CPU capability')[1][:40] if 'CPU capability' in torch.__config__.show() else '')
eng = make_engine('hip')
rel, stage, R, unit = True, 'color', 5000, False
(pos, geo, col, W), (dpos, dgeo, dcol, knn, dec) = T._gpu_scene(eng, 100_000, rel)
b = A.ray_batch(R, frame=7, holes=0.0, seed=1)
cfg = core.RenderCfg(rel_pos=rel)
st = core.RenderState(eng, R, cfg.S, need_act=True)
ro, rd, gd, gc = (eng.f32(b[k]) for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color'))
d_depth, d_color, out4 = eng.empty(R), eng.empty(R, 3), eng.zeros(4)
core.render_forward(eng, cfg, st, ro, rd, gd, knn, dpos, dgeo, dcol, dec, stage, save_act=True, extra_flags=_ffi.FLAG_ZERO_ABSENT, mapper_loss=(gc, 0.1, d_depth, d_color, out4))
torch.cuda.synchronize()
kn = T._check_knn_and_z(st, b, pos, 'dbg')
r0 = A.oracle_mapper(rel, stage, b, pos, geo, col, W, kn, grads=False)
bp, margin = A.branch_point_rays(r0['out'], b, pos, geo, W)
print('bp rays', int(bp.sum()), 'margin', margin)
d_depth[bp.to(eng.device)] = 0.0; d_color[bp.to(eng.device)] = 0.0
gs = core.GradState(eng, pos.shape[0], R, dec.n, feats=True, weights=True)
core.render_backward(eng, st, gs, d_depth, d_color)
torch.cuda.synchronize()
r = A.oracle_mapper(rel, stage, b, pos, geo, col, W, kn, exclude=bp)
got, ref = gs.g_col.cpu(), r['g_col']
err = (got - ref).abs()
scale = ref.abs().max()
rows = torch.nonzero(err.max(1).values > 1e-5 * scale).reshape(-1)
print('scale', float(scale), 'rows with err > 1e-5 scale:', rows.tolist()[:20], len(rows))
idx = torch.from_numpy(kn[1])
# d loss / d colour sign flips?  compare kernel d_color with oracle sign
o = r0['out']
dc_o = 0.1 * torch.sign(o['color'].detach() - b['gt_color']) * (o['valid_ray'] & (b['gt_depth'] > 0))[:, None]
dck = d_color.cpu()
bad = torch.nonzero(((dck - dc_o).abs().max(1).values > 1e-6) & ~bp).reshape(-1)
print('rays whose d colour differs (L1 kink of the colour term):', bad.tolist(), [(float(o['color'][i, c] - b['gt_color'][i, c])) for i in bad.tolist() for c in range(3)][:12])
for rw in rows.tolist()[:5]:
    smp = torch.nonzero((idx == rw).any(1)).reshape(-1)
    print('row', rw, 'err', float(err[rw].max() / scale), 'samples', smp.tolist()[:10], 'rays', sorted(set((smp // 5).tolist()))[:10])
d = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
b64 = {k: d(v) for k, v in b.items()}
r64 = A.oracle_mapper(rel, stage, b64, d(pos), d(geo), d(col), {k: d(v) for k, v in W.items()}, (kn[0].astype(np.float64), kn[1], kn[2]), exclude=bp)
for nm, g, r3, r6 in (('col_feats', gs.g_col.cpu(), r['g_col'], r64['g_col']), ('geo_feats', gs.g_geo.cpu(), r['g_geo'], r64['g_geo'])):
    sc = float(r6.abs().max())
    print(nm, 'gpu vs oracle32 %.2e | gpu vs oracle64 %.2e | oracle32 vs oracle64 %.2e' % (float((g - r3).abs().max()) / sc, float((g.double() - r6).abs().max()) / sc, float((r3.double() - r6).abs().max()) / sc))
gW = dec.unpack(gs.g_weights)
for name in ('color_decoder.pts_linears.1.weight', 'color_decoder.fc_c.2.weight', 'color_decoder.mlp_col_neighbor.linear1.weight', 'geo_decoder.embedder._B'):
    g, r3, r6 = gW[name].reshape(r['gW'][name].shape), r['gW'][name], r64['gW'][name]
    sc = float(r6.abs().max())
    print(name, 'gpu vs o32 %.2e | gpu vs o64 %.2e | o32 vs o64 %.2e' % (float((g - r3).abs().max()) / sc, float((g.double() - r6).abs().max()) / sc, float((r3.double() - r6).abs().max()) / sc))
