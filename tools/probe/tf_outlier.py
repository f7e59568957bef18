#!/usr/bin/env python3
"""Replay ONE optimize_map call recorded by tests/test_teacher_forced.py (LK_TF_DUMP) on the host emulator and against the oracle loop, iteration by iteration:
where do the stepped geometry rows of the product leave the oracle's?   python tools/probe/tf_outlier.py gpurun_out/tf_outlier.pt [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from loopy_slam_amd import config, core, slam, steps
import test_loops_at_size as L
import test_teacher_forced as TF
from util import make_engine

d = torch.load(sys.argv[1], weights_only=False)
rec, out = d['rec'], d['out']
iters = int(sys.argv[2]) if len(sys.argv) > 2 else rec['iters']
backend = sys.argv[3] if len(sys.argv) > 3 else 'emu'
cfg = TF._load('configs/TUM_RGBD/freiburg1_desk.yaml') if 'tum' in d['case'] else TF._load('configs/ScanNet/scene0000.yaml')
eng = make_engine(backend)
rcfg = slam.render_cfg_from(cfg, cfg['rendering']['sigmoid_coef_mapper'])
c = cfg['cam']; e = c.get('crop_edge', 0) or 0
Hh, Ww = c['H'] - 2 * e, c['W'] - 2 * e
intr = (c['fx'], c['fy'], c['cx'] - e, c['cy'] - e)
W, rows, N, R = rec['W'], rec['rows'], rec['pos'].shape[0], rec['R']
n_geo = min(iters, rec['geo_iters'] + 1)
print('case', d['case'], 'idx', rec['idx'], 'iters', rec['iters'], '->', iters, 'n_geo', n_geo, 'rows', rows.numel(), 'points', N, 'R', R, 'frames', rec['F'])
dpos = eng.f32(rec['pos'])
knn = core.KnnIndex(eng, capacity=N, cell_size=max(cfg['pointcloud']['radius_query'], 1e-3)); knn.build(dpos)
dec = core.DecoderBlob(eng).pack(W)
dgeo, dcol = eng.f32(rec['geo']).clone(), eng.f32(rec['col']).clone()
mo = steps.MapOptimizer(eng, rcfg, dec, knn, dpos, dgeo, dcol, None, R, rec['lrs'], w_color=rec['w_color'], dynamic_radius=rec['rstack'] is not None, fix_color_decoder=rec['fix_color_decoder'])
if 'statement' in sys.argv: mo.native_loop = False
mask = torch.zeros(N, dtype=torch.uint8); mask[rows] = 1
mo.new_frame(rows.to(torch.int32).to(eng.device), mask.to(eng.device))
log = eng.zeros(iters, 4)
frames = (eng.f32(rec['dstack']), eng.f32(rec['cstack']), eng.f32(rec['pstack']), eng.f32(rec['rstack']) if rec['rstack'] is not None else None)
mo.run(iters, n_geo, frames, rec['rnd'][:iters].to(torch.int32).to(eng.device), rec['fid'].to(torch.int32).to(eng.device), (0, Hh, 0, Ww), intr, Hh, Ww, log)
mo.finish()
kl = log[:, 0].cpu().numpy().astype(np.float64)
render = L.TreeRender(rec['pos'], rcfg.rel_pos, near=rcfg.near_surface, far=rcfg.far_surface, exact=('exact' in sys.argv))
render.cfg.radius_query, render.cfg.coef, render.cfg.min_nn = rcfg.radius_query, rcfg.coef, rcfg.min_nn
ol, og, oc, _ = L.oracle_map_loop(render, rec['geo'], rec['col'], W, rows, (rec['dstack'], rec['cstack'], rec['pstack']), rec['fid'], rec['rnd'][:iters], n_geo, intr, rec['lrs'],
                                  rec['dec_names'], w_color=rec['w_color'], rstack=rec['rstack'])
ol = np.array(ol)
print('loss rel', ['%.1e' % x for x in np.abs(kl - ol) / np.abs(ol)])
gk = dgeo.cpu()[rows]
err = (gk.double() - og.double()).abs()
moved = (og.double() - rec['geo'][rows].double()).abs()
st = L.param_error_stats(gk, og, rec['geo'][rows])
print('geo rows: q99 %.2e q999 %.2e max %.2e moved max %.2e' % (st['err_q99'], st['err_q999'], st['err_max'], st['moved_max']))
if 'product_geo_rows' in d and iters == rec['iters']:
    print('   emulator vs the recorded GPU run: q99 %.2e' % L.param_error_stats(gk, d['product_geo_rows'], rec['geo'][rows])['err_q99'])
# which rows?
row_err = err.max(dim=1).values
bad = torch.nonzero(row_err > 0.2 * rec['lrs']['geometry'][1]).reshape(-1)
print('rows with an entry more than 0.2 lr off:', bad.numel(), 'of', rows.numel(), '| entries off > 0.2 lr:', int((err > 0.2 * rec['lrs']['geometry'][1]).sum()))
torch.save({'bad': rows[bad], 'row_err': row_err, 'gk': gk, 'og': og}, '/tmp/tf_outlier_rows.pt')
print('oracle moved rows (any entry):', int((moved.max(dim=1).values > 0).sum()), ' product moved rows:', int(((gk.double() - rec['geo'][rows].double()).abs().max(dim=1).values > 0).sum()))
pm = (gk.double() - rec['geo'][rows].double()).abs().max(dim=1).values > 0
om = moved.max(dim=1).values > 0
print('moved in product only:', int((pm & ~om).sum()), ' oracle only:', int((om & ~pm).sum()))

if 'grads' in sys.argv:
    # the oracle's gradients of the stepped geometry rows, iteration by iteration: how large are the entries where the product ends elsewhere?
    from oracle import hotpath as H
    depth_s, color_s, pose_s = rec['dstack'], rec['cstack'], rec['pstack']
    F, Wd = depth_s.shape[0], depth_s.shape[2]
    fx, fy, cx, cy = intr
    geo_p = rec['geo'][rows].clone().requires_grad_(True)
    col_p = rec['col'][rows].clone().requires_grad_(True)
    dflat, cflat = depth_s.reshape(F, -1), color_s.reshape(F, -1, 3)
    gs = []
    opt = torch.optim.Adam([{'params': [geo_p], 'lr': 0}])
    for it in range(iters):
        stage = 'geometry' if it < n_geo else 'color'
        opt.param_groups[0]['lr'] = rec['lrs'][stage][1]
        opt.zero_grad()
        geo_t, col_t = rec['geo'].index_put((rows,), geo_p), rec['col'].index_put((rows,), col_p)
        fl = rec['rnd'][it].long()
        i, j = (fl % Wd).float(), torch.div(fl, Wd, rounding_mode='floor').float()
        dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
        Rm = pose_s[rec['fid']]
        rd = torch.sum(dirs[:, None, :] * Rm[:, :3, :3], -1); ro = Rm[:, :3, 3]
        gd, gc = dflat[rec['fid'], fl], cflat[rec['fid'], fl]
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        r2 = rec['rstack'].reshape(F, -1)[rec['fid'], fl][keep] if rec['rstack'] is not None else None
        o = render(ro[keep], rd[keep], gd[keep], geo_t, col_t, W, stage, r2_ray=r2)
        loss = H.mapper_loss(o['depth'], o['color'], o['valid_ray'], gd[keep], gc[keep], stage, rec['w_color'])[0]
        loss.backward()
        gs.append(geo_p.grad.detach().clone())
        opt.step()
    g_last = gs[-1].abs()
    big = err > 0.2 * rec['lrs']['geometry'][1]
    print('oracle |g| of the LAST iteration: all entries with a gradient: median %.2e, q10 %.2e, q90 %.2e, max %.2e' % tuple(float(torch.quantile(g_last[g_last > 0].float()[:4000000], q)) for q in (0.5, 0.1, 0.9, 1.0)))
    gb = g_last[big]
    print('   entries where the product ends > 0.2 lr elsewhere (%d): |g| median %.2e, q10 %.2e, q90 %.2e, max %.2e, zero %d' % ((int(big.sum()),) + tuple(float(torch.quantile(gb.float(), q)) for q in (0.5, 0.1, 0.9, 1.0)) + (int((gb == 0).sum()),)))
    for lo, hi in ((0, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1)):
        sel = (g_last > lo) & (g_last <= hi)
        print('   |g| in (%.0e, %.0e]: %8d entries, %6d of them off > 0.2 lr (%.2f %%)' % (lo, hi, int(sel.sum()), int((sel & big).sum()), 100.0 * float((sel & big).sum()) / max(1, int(sel.sum()))))

Wk = dec.unpack()
_, _, _, Wo = L.oracle_map_loop(render, rec['geo'], rec['col'], W, rows, (rec['dstack'], rec['cstack'], rec['pstack']), rec['fid'], rec['rnd'][:iters], n_geo, intr, rec['lrs'],
                                rec['dec_names'], w_color=rec['w_color'], rstack=rec['rstack']) if 'dec' in sys.argv else (0, 0, 0, None)
if Wo is not None:
    n = 'geo_decoder.embedder._B'
    a, b, c = Wk[n].reshape(Wo[n].shape), Wo[n], W[n]
    print(n, 'product step / lr:', ((a - c) / rec['lrs']['geometry'][0]).reshape(-1)[:96].round(decimals=2).tolist())
    print(n, 'oracle  step / lr:', ((b - c) / rec['lrs']['geometry'][0]).reshape(-1)[:96].round(decimals=2).tolist())
    print('entries that differ by more than 0.5 lr:', int(((a - b).abs() > 0.5 * rec['lrs']['geometry'][0]).sum()), 'of', a.numel())
