"""End-to-end parity of the per-frame optimisation loops (loopy_slam_amd.steps) against an oracle loop
written with torch autograd + torch.optim.Adam over the oracle's render (oracle/hotpath.py), i.e. the
reference's Mapper.optimize_map / Tracker.optimize_cam_in_batch inner loops on a miniature scene."""
import math

import numpy as np
import pytest
import torch

from oracle import hotpath as H
from loopy_slam_amd import core, steps, synthetic as syn
from util import make_engine, backends, relerr

torch.set_num_threads(1)
HH, WW = 24, 32
INTR = (40.0, 40.0, 15.5, 11.5)


def mini_scene(seed=0):
    g = torch.Generator().manual_seed(seed)
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([0.1, -0.05, 0.2])
    jj, ii = torch.meshgrid(torch.arange(HH, dtype=torch.float32), torch.arange(WW, dtype=torch.float32), indexing='ij')
    ro, rd = H.rays_from_uv(ii.reshape(-1), jj.reshape(-1), c2w, *INTR)
    depth = (1.0 + 0.15 * torch.sin(ii / 5.0) * torch.cos(jj / 4.0)).reshape(-1)
    depth[::17] = 0.0                                   # holes
    depth[5] = 40.0                                     # outlier beyond the inside mask
    color = torch.rand(HH * WW, 3, generator=g)
    pts = []
    ok = (depth > 0) & (depth < 10)
    for t in (0.98, 1.0, 1.02):
        pts.append((ro + rd * (depth * t)[:, None])[ok] + 0.004 * torch.randn(int(ok.sum()), 3, generator=g))
    pos = torch.cat(pts).float().contiguous()
    geo = (0.1 * torch.randn(pos.shape[0], 32, generator=g)).float()
    col = (0.1 * torch.randn(pos.shape[0], 32, generator=g)).float()
    return c2w, depth.reshape(HH, WW), color.reshape(HH, WW, 3), pos, geo, col


def oracle_rays(c2w, depth_img, color_img, rnd):
    i = (rnd % WW).float()
    j = (rnd // WW).float()
    ro, rd = H.rays_from_uv(i, j, c2w, *INTR)
    gd = depth_img.reshape(-1)[rnd.long()]
    gc = color_img.reshape(-1, 3)[rnd.long()]
    return ro, rd, gd, gc, i, j


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('native,frozen,all_rows,geo_free', ((True, False, False, False), (False, False, False, False), (True, True, False, False),
                                                              (True, True, True, False), (True, False, False, True)))
@pytest.mark.parametrize('rel_pos', (True, False))
def test_map_iterations_match_oracle(backend, rel_pos, native, frozen, all_rows, geo_free):
    """native: the whole loop as ONE lk_map_frame call (MapOptimizer.run); else one launch sequence per statement (iterate).
    geo_free: mapping.fix_geo_decoder False (Mapper.py:524-526) - the geometry decoder's matrices are trained too (run() then takes
    the per-statement path by itself).
    all_rows: the whole-map refinement - rows = NULL, every row a parameter; lk_map_frame then steps only the rows its gathers have flagged
    (lk_adam_seg::row_flags), which must equal torch.optim.Adam over the whole tables."""
    eng = make_engine(backend)
    c2w, depth_img, color_img, pos, geo, col = mini_scene()
    W = syn.default_weights(seed=7)
    R, iters = 96, 3
    g = torch.Generator().manual_seed(11)
    rnd_all = torch.randint(0, HH * WW, (iters, R), generator=g, dtype=torch.int32)
    rows = torch.arange(0, pos.shape[0], 1 if all_rows else 2, dtype=torch.int32)               # "frustum" = every other point
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    stages = ['geometry', 'color', 'color']
    # ---------------- oracle loop (reference semantics: params = clones of the selected rows)
    ocfg = H.RenderCfg(rel_pos=rel_pos)
    Wt = {k: v.clone() for k, v in W.items()}
    # frozen = fix_color_decoder (the end-of-sequence refinement, Mapper.py:531-541): of the decoders only the Fourier matrices stay
    # trainable - lk_map_frame then renders its backward with LK_FLAG_EMBED_GRADS_ONLY (no weight-gradient rows, no reduction launches)
    dec_names = list(steps.GEO_DECODER_ALL_PARAMS if geo_free else steps.GEO_DECODER_PARAMS) + \
        ([n for n in steps.COLOR_DECODER_PARAMS] if not frozen else (['color_decoder.embedder_rel_pos._B'] if rel_pos else []))
    for n in dec_names:
        Wt[n].requires_grad_(True)
    geo_o, col_o = geo.clone(), col.clone()
    geo_p = geo_o[rows.long()].clone().requires_grad_(True)
    col_p = col_o[rows.long()].clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [Wt[n] for n in dec_names], 'lr': 0}, {'params': [geo_p], 'lr': 0}, {'params': [col_p], 'lr': 0}])
    o_losses = []
    for it in range(iters):
        stage = stages[it]
        for gi in range(3):
            opt.param_groups[gi]['lr'] = lrs[stage][gi]
        opt.zero_grad()
        geo_t = geo_o.clone(); geo_t[rows.long()] = geo_p
        col_t = col_o.clone(); col_t[rows.long()] = col_p
        ro, rd, gd, gc, _, _ = oracle_rays(c2w, depth_img, color_img, rnd_all[it])
        keep = gd > 0
        thr = H.inside_threshold(gd[keep])
        keep = keep & (gd <= thr)
        out = H.render_batch(ocfg, ro[keep], rd[keep], gd[keep], pos, geo_t, col_t, Wt, stage)
        loss, _, _, _ = H.mapper_loss(out['depth'], out['color'], out['valid_ray'], gd[keep], gc[keep], stage, 0.1)
        loss.backward()
        opt.step()
        o_losses.append(loss.item())
    # ---------------- kernels
    cfg = core.RenderCfg(rel_pos=rel_pos)
    dec = core.DecoderBlob(eng).pack(W)
    pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
    knn = core.KnnIndex(eng, capacity=pos.shape[0])
    knn.build(pos_d)
    mo = steps.MapOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, None if all_rows else rows.to(eng.device), R, lrs, w_color=0.1,
                            fix_color_decoder=frozen, fix_geo_decoder=not geo_free)
    mo.begin_frame()
    frames = (eng.f32(depth_img).reshape(1, HH, WW), eng.f32(color_img).reshape(1, HH, WW, 3), eng.f32(c2w).reshape(1, 4, 4), None)
    fid = torch.zeros(R, dtype=torch.int32, device=eng.device)
    k_losses = []
    if native:
        log = eng.zeros(iters, 4)
        mo.run(iters, 1, frames, rnd_all.to(eng.device), fid, (0, HH, 0, WW), INTR, HH, WW, log)
        k_losses = [float(x) for x in log[:, 0].cpu()]
    else:
        for it in range(iters):
            out4 = mo.iterate(stages[it], frames, rnd_all[it].to(eng.device), fid, (0, HH, 0, WW), INTR, HH, WW)
            k_losses.append(float(out4[0].cpu()))
    np.testing.assert_allclose(k_losses, o_losses, rtol=2e-4)
    r = rows.long()
    # selected rows moved by Adam, the others untouched
    assert float((geo_d.cpu()[r] - geo[r]).abs().max()) > 1e-3
    if all_rows:        # rows no batch touched: exactly where they were (zero gradient, zero moments), on both sides
        still = (geo_p.detach() == geo).all(1) & (col_p.detach() == col).all(1)
        assert 0 < int(still.sum()) < still.numel()
        assert torch.equal(geo_d.cpu()[still], geo[still]) and torch.equal(col_d.cpu()[still], col[still])
    other = torch.ones(pos.shape[0], dtype=torch.bool); other[r] = False
    assert torch.equal(geo_d.cpu()[other], geo[other]) and torch.equal(col_d.cpu()[other], col[other])
    # Adam's first steps are sign-like (lr * g / (|g| + 1e-8)): entries whose gradient is of the order of eps
    # amplify fp32 summation-order noise, so bound the bulk tightly and the tail by a fraction of one lr step
    for mine, ref, lr in ((geo_d.cpu()[r], geo_p.detach(), 0.03), (col_d.cpu()[r], col_p.detach(), 0.005)):
        err = (mine - ref).abs().reshape(-1)
        # (three levels: 99 % of the entries to 2e-5, 99.9 % to 2 % of one lr step; an entry whose gradient sign is decided by
        # rounding noise in every iteration can end up to 2 lr per iteration away, which is Adam's hard limit)
        assert float(torch.quantile(err, 0.99)) < 2e-5, float(torch.quantile(err, 0.99))
        assert float(torch.quantile(err, 0.999)) < 0.02 * lr, float(torch.quantile(err, 0.999))
        assert float(err.max()) < 2.0 * lr * iters, float(err.max())
    Wk = dec.unpack()
    if frozen:          # the frozen matrices are bit for bit what they were
        for n in steps.COLOR_DECODER_PARAMS:
            if n in Wk and n not in dec_names:
                assert torch.equal(Wk[n].reshape(W[n].shape), W[n]), n
    for n in dec_names:
        if n not in Wk:
            continue
        if not rel_pos and ('mlp_col_neighbor' in n or 'embedder_rel_pos' in n):
            continue
        err = (Wk[n].reshape(Wt[n].shape) - Wt[n].detach()).abs().reshape(-1)
        moved = float((Wt[n].detach() - W[n]).abs().max())
        # same bounds as the feature rows: bulk tight, sign-like tail below Adam's hard limit (twice what a weight can move)
        bulk = float(torch.quantile(err, 0.999)) if err.numel() > 1000 else float(err.max())
        assert bulk <= 2e-4 * max(1.0, float(W[n].abs().max())) + 0.05 * moved, (n, bulk, moved)
        assert float(err.max()) <= 2e-4 * max(1.0, float(W[n].abs().max())) + 2.0 * moved, (n, float(err.max()), moved)


@pytest.mark.gpu
def test_long_map_call_matches_the_statement_path():
    """130 iterations in ONE lk_map_frame call: the look-ahead chunks are then nine iterations long - more than one batch of the
    batched row sort (LK_SEG_BATCH = 8), rows counted by k_seg_count instead of inside the search - and the third stream runs
    fourteen chunks ahead of the loop.  Reference: the same iterations as one launch sequence per statement (MapOptimizer.iterate:
    search, count and sort inside every iteration).  Small learning rates keep the two fp32 trajectories together."""
    eng = make_engine('hip')
    c2w, depth_img, color_img, pos, geo, col = mini_scene()
    W = syn.default_weights(seed=7)
    R, iters, n_geo = 96, 130, 40
    g = torch.Generator().manual_seed(5)
    rnd_all = torch.randint(0, HH * WW, (iters, R), generator=g, dtype=torch.int32).to(eng.device)
    rows = torch.arange(0, pos.shape[0], 2, dtype=torch.int32).to(eng.device)
    lrs = {'geometry': (1e-4, 1e-3, 0.0), 'color': (2e-4, 2e-4, 2e-4)}
    frames = (eng.f32(depth_img).reshape(1, HH, WW), eng.f32(color_img).reshape(1, HH, WW, 3), eng.f32(c2w).reshape(1, 4, 4), None)
    fid = torch.zeros(R, dtype=torch.int32, device=eng.device)
    res = {}
    for native in (True, False):
        cfg = core.RenderCfg(rel_pos=True)
        dec = core.DecoderBlob(eng).pack(W)
        pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
        knn = core.KnnIndex(eng, capacity=pos.shape[0])
        knn.build(pos_d)
        mo = steps.MapOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, rows, R, lrs, w_color=0.1)
        mo.begin_frame()
        if native:
            log = eng.zeros(iters, 4)
            mo.run(iters, n_geo, frames, rnd_all, fid, (0, HH, 0, WW), INTR, HH, WW, log)
            losses = log[:, 0].cpu().numpy().copy()
        else:
            losses = np.array([float(mo.iterate('geometry' if it < n_geo else 'color', frames, rnd_all[it], fid, (0, HH, 0, WW), INTR, HH, WW)[0].cpu())
                               for it in range(iters)])
        res[native] = (losses, geo_d.cpu().clone(), col_d.cpu().clone(), {k: v.clone() for k, v in dec.unpack().items()})
    la, lb = res[True][0], res[False][0]
    assert np.isfinite(la).all() and la[-1] != la[n_geo]
    np.testing.assert_allclose(la, lb, rtol=2e-3)
    for i in (1, 2):
        err = (res[True][i] - res[False][i]).abs()
        assert float(torch.quantile(err.reshape(-1), 0.999)) < 2e-4 and float(err.max()) < 0.02, (i, float(err.max()))
    for n, w in res[True][3].items():
        assert float((w - res[False][3][n]).abs().max()) <= 2e-3 * max(1.0, float(w.abs().max())), n


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('exposure', (False, True))
def test_prepared_map_call_equals_the_plain_one(backend, exposure):
    """MapOptimizer.prepare (lk_map_prepare: gradient-table fills + the batch assembly of the call enqueued BEFORE the row selection's count
    read-back) followed by new_frame(zero=False) + run gives what new_frame + run gives; a prepare whose arguments do not match the run
    (other draws) is ignored."""
    eng = make_engine(backend)
    c2w, depth_img, color_img, pos, geo, col = mini_scene(6)
    W = syn.default_weights(seed=3, rel_pos=not exposure, exposure=exposure)
    R, iters, n_geo = 64, 4, 2
    g = torch.Generator().manual_seed(44)
    rnd_all = torch.randint(0, HH * WW, (iters, R), generator=g, dtype=torch.int32).to(eng.device)
    rnd_other = torch.randint(0, HH * WW, (iters, R), generator=g, dtype=torch.int32).to(eng.device)
    rows = torch.arange(0, pos.shape[0], 2, dtype=torch.int32).to(eng.device)
    mask = torch.zeros(pos.shape[0], dtype=torch.uint8); mask[::2] = 1
    mask = mask.to(eng.device)
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    frames = (eng.f32(depth_img).reshape(1, HH, WW), eng.f32(color_img).reshape(1, HH, WW, 3), eng.f32(c2w).reshape(1, 4, 4), None)
    fid = torch.zeros(R, dtype=torch.int32, device=eng.device)
    res = {}
    for mode in ('plain', 'prepared', 'stale'):
        cfg = core.RenderCfg(rel_pos=not exposure, exposure=exposure)
        dec = core.DecoderBlob(eng).pack(W)
        pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
        knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(pos_d)
        xp = None
        if exposure:
            xp = (_exposure_module(W).to(eng.device), [(0.2 * torch.ones(8)).to(eng.device).requires_grad_(True)])
        mo = steps.MapOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, None, R, lrs, w_color=0.1, exposure=xp)
        log = eng.zeros(iters, 4)
        args = (iters, n_geo, frames, rnd_all, fid, (0, HH, 0, WW), INTR, HH, WW, log)
        if mode == 'plain':
            mo.new_frame(rows, mask)
        else:
            ok = mo.prepare(*(args if mode == 'prepared' else (iters, n_geo, frames, rnd_other, fid, (0, HH, 0, WW), INTR, HH, WW, log)))
            assert ok
            mo.new_frame(rows, mask, zero=False)
        mo.run(*args)
        mo.finish()
        res[mode] = (log.cpu().clone(), geo_d.cpu().clone(), col_d.cpu().clone(), dec.blob.cpu().clone())
    for mode in ('prepared', 'stale'):
        np.testing.assert_allclose(res[mode][0].numpy(), res['plain'][0].numpy(), rtol=1e-6, atol=1e-7)
        for k in (1, 2, 3):
            err = (res[mode][k] - res['plain'][k]).abs().reshape(-1)
            assert float(torch.quantile(err, 0.999)) < 2e-6 and float(err.max()) < 0.02 * 0.03, (mode, k, float(err.max()))
    assert float((res['plain'][1] - geo).abs().max()) > 1e-3


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('exposure', (False, True))
def test_segmented_map_call_equals_the_unsegmented_one(backend, exposure):
    """A call longer than MapOptimizer.max_call_iters is issued as consecutive lk_map_frame segments (lk_map_desc::it_offset: the work buffer
    holds one segment's batches and lists): seven iterations, the stage change inside the second segment, in segments of three against one
    call - Adam's step counts and moments carry over, the losses and every parameter agree."""
    eng = make_engine(backend)
    c2w, depth_img, color_img, pos, geo, col = mini_scene(4)
    W = syn.default_weights(seed=5, rel_pos=not exposure, exposure=exposure)
    R, iters, n_geo = 64, 7, 4
    g = torch.Generator().manual_seed(33)
    rnd_all = torch.randint(0, HH * WW, (iters, R), generator=g, dtype=torch.int32).to(eng.device)
    rows = torch.arange(0, pos.shape[0], 2, dtype=torch.int32).to(eng.device)
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    frames = (eng.f32(depth_img).reshape(1, HH, WW), eng.f32(color_img).reshape(1, HH, WW, 3), eng.f32(c2w).reshape(1, 4, 4), None)
    fid = torch.zeros(R, dtype=torch.int32, device=eng.device)
    res = {}
    for seg in (128, 3):
        cfg = core.RenderCfg(rel_pos=not exposure, exposure=exposure)
        dec = core.DecoderBlob(eng).pack(W)
        pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
        knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(pos_d)
        xp = None
        if exposure:
            mlp = _exposure_module(W).to(eng.device)
            feat = (0.2 * torch.ones(8)).to(eng.device).requires_grad_(True)
            xp = (mlp, [feat])
        mo = steps.MapOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, rows, R, lrs, w_color=0.1, exposure=xp)
        mo.max_call_iters = seg
        mo.begin_frame()
        log = eng.zeros(iters, 4)
        mo.run(iters, n_geo, frames, rnd_all, fid, (0, HH, 0, WW), INTR, HH, WW, log)
        mo.finish()
        res[seg] = (log.cpu().clone(), geo_d.cpu().clone(), col_d.cpu().clone(), dec.blob.cpu().clone(),
                    feat.detach().cpu().clone() if exposure else None)
    a, b = res[128], res[3]
    np.testing.assert_allclose(b[0].numpy(), a[0].numpy(), rtol=1e-6, atol=1e-7)
    assert float((a[1] - geo).abs().max()) > 1e-3                                     # the rows did move
    # (on the chip two runs of the SAME loop differ in a handful of entries: float atomics order + Adam's sign-like first steps on entries
    # whose gradient is of the order of eps - the bounds of the other loop tests: bulk tight, tail below a fraction of one lr step)
    for k in (1, 2, 3):
        err = (b[k] - a[k]).abs().reshape(-1)
        assert float(torch.quantile(err, 0.999)) < 2e-6 and float(err.max()) < 0.02 * 0.03, (k, float(torch.quantile(err, 0.999)), float(err.max()))
    if exposure:
        np.testing.assert_allclose(b[4].numpy(), a[4].numpy(), rtol=0, atol=2e-6)
        assert float((a[4] - 0.2).abs().max()) > 1e-4


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('rel_pos', (True, False))
def test_map_iterations_with_ba_match_oracle(backend, rel_pos):
    """mapping.BA (Mapper.py:541-566, 602-607, 629-643, 685, 782-797): three frames in the window, the oldest fixed, the other two
    poses optimised as 7-vectors in one more Adam group whose lr is non-zero in a window of the colour iterations only (moments
    accumulate from the first iteration on, as torch.optim.Adam does at lr 0); the batch is rendered with is_tracker=True.  Losses,
    poses, feature rows and decoder weights against an autograd + torch.optim.Adam loop over the oracle's render."""
    eng = make_engine(backend)
    c2w, depth_img, color_img, pos, geo, col = mini_scene(3)
    W = syn.default_weights(seed=9)
    F, n, iters = 3, 40, 4
    R = F * n
    g = torch.Generator().manual_seed(21)
    rnd_all = torch.randint(0, HH * WW, (iters, R), generator=g, dtype=torch.int32)
    rows = torch.arange(0, pos.shape[0], 2, dtype=torch.int32)
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    stages = ['geometry', 'color', 'color', 'color']
    cam_lr = lambda it: 0.002 if it in (1, 2) else 0.0
    cam_c = H.c2w_to_cam(c2w)
    offs = [torch.zeros(7), torch.tensor([0.0, 0.003, -0.002, 0.001, 0.004, -0.002, 0.003]), torch.tensor([0.0, -0.002, 0.002, 0.002, -0.003, 0.004, -0.002])]
    cams0 = torch.stack([cam_c + o for o in offs])
    c2w_stack = torch.stack([torch.cat([H.quat_to_c2w(c), torch.tensor([[0.0, 0.0, 0.0, 1.0]])]) for c in cams0])
    trainable = [False, True, True]
    # ---------------- oracle loop
    ocfg = H.RenderCfg(rel_pos=rel_pos)
    Wt = {k: v.clone() for k, v in W.items()}
    dec_names = list(steps.GEO_DECODER_PARAMS) + [nm for nm in steps.COLOR_DECODER_PARAMS if nm in Wt and (rel_pos or ('mlp_col_neighbor' not in nm and 'embedder_rel_pos' not in nm))]
    for nm in dec_names:
        Wt[nm].requires_grad_(True)
    geo_p, col_p = geo[rows.long()].clone().requires_grad_(True), col[rows.long()].clone().requires_grad_(True)
    cam_p = [cams0[f].clone().requires_grad_(True) for f in range(F) if trainable[f]]
    opt = torch.optim.Adam([{'params': [Wt[nm] for nm in dec_names], 'lr': 0}, {'params': [geo_p], 'lr': 0}, {'params': [col_p], 'lr': 0},
                            {'params': cam_p, 'lr': 0}])
    o_losses = []
    for it in range(iters):
        stage = stages[it]
        for gi in range(3):
            opt.param_groups[gi]['lr'] = lrs[stage][gi]
        opt.param_groups[3]['lr'] = cam_lr(it)
        opt.zero_grad()
        geo_t = geo.clone(); geo_t[rows.long()] = geo_p
        col_t = col.clone(); col_t[rows.long()] = col_p
        ros, rds, gds, gcs = [], [], [], []
        k = 0
        for f in range(F):
            rr = rnd_all[it, f * n:(f + 1) * n]
            if trainable[f]:
                pose = H.quat_to_c2w(cam_p[k]); k += 1
            else:
                pose = c2w_stack[f]
            ro, rd = H.rays_from_uv((rr % WW).float(), (rr // WW).float(), pose, *INTR)
            ros.append(ro); rds.append(rd)
            gds.append(depth_img.reshape(-1)[rr.long()]); gcs.append(color_img.reshape(-1, 3)[rr.long()])
        ro, rd, gd, gc = torch.cat(ros), torch.cat(rds), torch.cat(gds), torch.cat(gcs)
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        out = H.render_batch(ocfg, ro[keep], rd[keep], gd[keep], pos, geo_t, col_t, Wt, stage, tracker=True)
        loss, _, _, _ = H.mapper_loss(out['depth'], out['color'], out['valid_ray'], gd[keep], gc[keep], stage, 0.1)
        loss.backward()
        opt.step()
        o_losses.append(loss.item())
    # ---------------- kernels (per-statement path: BA is not part of lk_map_frame)
    cfg = core.RenderCfg(rel_pos=rel_pos)
    dec = core.DecoderBlob(eng).pack(W)
    pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
    knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(pos_d)
    mo = steps.MapOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, rows.to(eng.device), R, lrs, w_color=0.1)
    mo.begin_frame()
    cams_d = eng.f32(cams0).clone().contiguous()
    mo.enable_ba(cams_d, trainable, cam_lr)
    frames = (eng.f32(depth_img).reshape(1, HH, WW).repeat(F, 1, 1).contiguous(), eng.f32(color_img).reshape(1, HH, WW, 3).repeat(F, 1, 1, 1).contiguous(),
              eng.f32(c2w_stack).contiguous(), None)
    fid = torch.arange(F, dtype=torch.int32).repeat_interleave(n).to(eng.device)
    log = eng.zeros(iters, 4)
    mo.run(iters, 1, frames, rnd_all.to(eng.device), fid, (0, HH, 0, WW), INTR, HH, WW, log)
    np.testing.assert_allclose(log[:, 0].cpu().numpy(), o_losses, rtol=2e-4)
    cams_k = cams_d.cpu()
    assert torch.equal(cams_k[0], cams0[0])                                   # the oldest frame is fixed
    k = 0
    for f in range(F):
        if trainable[f]:
            moved = float((cam_p[k].detach() - cams0[f]).abs().max())
            assert moved > 1e-3                                               # two steps of 2e-3 (Adam's first steps are sign-like)
            np.testing.assert_allclose(cams_k[f].numpy(), cam_p[k].detach().numpy(), rtol=0, atol=2e-5)
            k += 1
    r = rows.long()
    for mine, ref in ((geo_d.cpu()[r], geo_p.detach()), (col_d.cpu()[r], col_p.detach())):
        err = (mine - ref).abs().reshape(-1)
        assert float(torch.quantile(err, 0.99)) < 5e-5 and float(err.max()) < 0.03, (float(torch.quantile(err, 0.99)), float(err.max()))
    Wk = dec.unpack()
    for nm in dec_names:
        # bulk tight; the tail by what the weight moved: an entry whose gradient is rounding noise in one iteration takes Adam's sign-like
        # step the other way (the bounds of test_map_iterations_match_oracle)
        err = (Wk[nm].reshape(Wt[nm].shape) - Wt[nm].detach()).abs().reshape(-1)
        scale, moved = max(1.0, float(Wt[nm].detach().abs().max())), float((Wt[nm].detach() - W[nm]).abs().max())
        bulk = float(torch.quantile(err, 0.999)) if err.numel() > 1000 else float(err.max())
        assert bulk <= 2e-3 * scale and float(err.max()) <= 2e-3 * scale + 2.0 * moved, (nm, bulk, float(err.max()), moved)


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('native,separate,handle_dynamic', ((True, True, True), (False, True, True), (True, False, True), (False, False, True),
                                                            (True, True, False), (False, False, False)))
def test_track_iterations_match_oracle(backend, native, separate, handle_dynamic):
    """native: the loop as ONE lk_track_frame call (fused one-workgroup kernels for batch assembly / loss / pose update);
    separate: tracking.separate_LR - two Adam groups and the candidate pose taken BEFORE the step (Replica); otherwise one
    leaf tensor stepped in place, the candidate is the pose AFTER the step (TUM / ScanNet; Tracker.py:334-377);
    handle_dynamic False: the median-of-residual outlier mask (Tracker.py:177-179; a few depth pixels are pushed out so that it bites)."""
    eng = make_engine(backend)
    c2w, depth_img, color_img, pos, geo, col = mini_scene(1)
    W = syn.default_weights(seed=8)
    R, iters, lr = 80, 4, 0.002
    g = torch.Generator().manual_seed(12)
    win = (2, HH - 2, 2, WW - 2)
    w_w = win[3] - win[2]
    n_win = (win[1] - win[0]) * w_w
    rnd_all = torch.randint(0, n_win, (iters, R), generator=g, dtype=torch.int32)
    cam0 = H.c2w_to_cam(c2w) + torch.tensor([0.0, 0.002, -0.001, 0.0015, 0.004, -0.003, 0.002])
    # ---------------- oracle loop (Tracker.py:313-401, separate_LR)
    ocfg = H.RenderCfg(rel_pos=True)
    q = cam0[:4].clone().requires_grad_(True)
    T = cam0[4:].clone().requires_grad_(True)
    cam_leaf = cam0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [T], 'lr': lr}, {'params': [q], 'lr': 0.2 * lr}]) if separate else \
        torch.optim.Adam([{'params': [cam_leaf], 'lr': lr}])
    o_losses, o_cams, o_masked = [], [], []
    if not handle_dynamic:          # depth outliers (a moving object in front of the wall) that 10 x the median residual rejects
        gen_o = torch.Generator().manual_seed(5)
        hit = torch.rand(depth_img.shape, generator=gen_o) < 0.04
        depth_img = torch.where(hit & (depth_img > 0), depth_img * 0.8, depth_img)
    for it in range(iters):
        cam = torch.cat([q, T]) if separate else cam_leaf
        if separate:
            o_cams.append(cam.detach().clone())
        opt.zero_grad()
        rr = rnd_all[it]
        i = (win[2] + rr % w_w).float()
        j = (win[0] + rr // w_w).float()
        ro, rd = H.rays_from_uv(i, j, H.quat_to_c2w(cam), *INTR)
        flat = (j.long() * WW + i.long())
        gd = depth_img.reshape(-1)[flat]
        gc = color_img.reshape(-1, 3)[flat]
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        out = H.render_batch(ocfg, ro[keep], rd[keep], gd[keep], pos, geo, col, W, 'color', tracker=True)
        loss, _, _, m_o = H.tracker_loss(out['depth'], out['var'], out['color'], gd[keep], gc[keep], 0.5, handle_dynamic=handle_dynamic)
        o_masked.append(int(m_o.sum()))
        loss.backward()
        opt.step()
        if not separate:
            o_cams.append(cam_leaf.detach().clone())
        o_losses.append(loss.item())
    # ---------------- kernels
    cfg = core.RenderCfg(rel_pos=True)
    dec = core.DecoderBlob(eng).pack(W)
    pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo), eng.f32(col)
    knn = core.KnnIndex(eng, capacity=pos.shape[0])
    knn.build(pos_d)
    to = steps.TrackOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, R, lr, separate_lr=separate, w_color=0.5, handle_dynamic=handle_dynamic)
    to.native_loop = native
    best, log = to.track(eng.f32(cam0), eng.f32(depth_img), eng.f32(color_img), iters, win, INTR, rnd_all.to(eng.device))
    np.testing.assert_allclose(log[:, 0].cpu().numpy(), o_losses, rtol=5e-4)
    assert log[:, 3].cpu().numpy().astype(int).tolist() == o_masked
    if not handle_dynamic:
        assert min(o_masked) < R - 2          # the mask did reject rays
    k = int(np.argmin(o_losses))
    np.testing.assert_allclose(best.cpu().numpy(), o_cams[k].numpy(), rtol=0, atol=2e-5)


def _exposure_module(W):
    """torch MLP_exposure (8 -> 128 softplus100 -> 12) holding the oracle weights."""
    m = torch.nn.Sequential(torch.nn.Linear(8, 128), torch.nn.Softplus(beta=100), torch.nn.Linear(128, 12))
    with torch.no_grad():
        m[0].weight.copy_(W['color_decoder.mlp_exposure.linear1.weight']); m[0].bias.copy_(W['color_decoder.mlp_exposure.linear1.bias'])
        m[2].weight.copy_(W['color_decoder.mlp_exposure.linear2.weight']); m[2].bias.copy_(W['color_decoder.mlp_exposure.linear2.bias'])
    return m


@pytest.mark.parametrize('native', (True, False))
@pytest.mark.parametrize('backend', backends())
def test_exposure_iterations_match_oracle(backend, native):
    """model.encode_exposure (ScanNet): tracker iterations with the decoder-side per-sample affine of the frame's
    exposure feature (Tracker.py:329-344), then mapper colour iterations with the per-keyframe affine on the rendered
    logits (Mapper.py:697-715) - exposure features and MLP optimised at lr 1e-3, against oracle loops.  native: the loops as
    single C-ABI calls (lk_track_frame / lk_map_frame with an lk_exposure_desc), else the per-statement path."""
    eng = make_engine(backend)
    c2w, depth_img, color_img, pos, geo, col = mini_scene(2)
    W = syn.default_weights(seed=4, rel_pos=False, exposure=True)
    gen = torch.Generator().manual_seed(31)
    # ------------------------------------------------------------------ tracker
    R, iters, lr = 72, 3, 0.002
    win = (2, HH - 2, 2, WW - 2)
    w_w = win[3] - win[2]
    rnd_all = torch.randint(0, (win[1] - win[0]) * w_w, (iters, R), generator=gen, dtype=torch.int32)
    cam0 = H.c2w_to_cam(c2w) + torch.tensor([0.0, 0.002, -0.001, 0.0015, 0.004, -0.003, 0.002])
    feat0 = 0.3 * torch.randn(8, generator=gen)
    ocfg = H.RenderCfg(rel_pos=False, exposure=True)
    Wo = {k: (v.clone().requires_grad_(True) if 'mlp_exposure' in k else v) for k, v in W.items()}
    cam = cam0.clone().requires_grad_(True)
    feat = feat0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [cam], 'lr': lr}, {'params': [feat], 'lr': 0.001},
                            {'params': [v for k, v in Wo.items() if 'mlp_exposure' in k], 'lr': 0.001}])
    o_losses = []
    for it in range(iters):
        opt.zero_grad()
        rr = rnd_all[it]
        i, j = (win[2] + rr % w_w).float(), (win[0] + rr // w_w).float()
        ro, rd = H.rays_from_uv(i, j, H.quat_to_c2w(cam), *INTR)
        flat = j.long() * WW + i.long()
        gd, gc = depth_img.reshape(-1)[flat], color_img.reshape(-1, 3)[flat]
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        out = H.render_batch(ocfg, ro[keep], rd[keep], gd[keep], pos, geo, col, Wo, 'color', tracker=True,
                             affine=H.exposure_affine(Wo, feat))
        loss, _, _, _ = H.tracker_loss(out['depth'], out['var'], out['color'], gd[keep], gc[keep], 0.5)
        loss.backward()
        opt.step()
        o_losses.append(loss.item())
    cfg = core.RenderCfg(rel_pos=False, exposure=True)
    dec = core.DecoderBlob(eng).pack(W)
    pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo), eng.f32(col)
    knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(pos_d)
    mlp = _exposure_module(W).to(eng.device)
    feat_k = eng.f32(feat0).clone().requires_grad_(True)
    to = steps.TrackOptimizer(eng, cfg, dec, knn, pos_d, geo_d, col_d, R, lr, separate_lr=False, w_color=0.5)
    to.native_loop = native
    best, log = to.track(eng.f32(cam0), eng.f32(depth_img), eng.f32(color_img), iters, win, INTR, rnd_all.to(eng.device),
                         exposure=(mlp, feat_k))
    np.testing.assert_allclose(log[:, 0].cpu().numpy(), o_losses, rtol=5e-4)
    np.testing.assert_allclose(feat_k.detach().cpu().numpy(), feat.detach().numpy(), atol=2e-4)
    assert float((feat_k.detach().cpu() - feat0).abs().max()) > 5e-4                       # the feature did move
    np.testing.assert_allclose(mlp[2].bias.detach().cpu().numpy(), Wo['color_decoder.mlp_exposure.linear2.bias'].detach().numpy(), atol=2e-4)
    # ------------------------------------------------------------------ mapper, colour stage, two keyframes
    Rm, iters_m = 96, 2
    rows = torch.arange(0, pos.shape[0], 2, dtype=torch.int32)
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    rnd_m = torch.randint(0, HH * WW, (iters_m, Rm), generator=gen, dtype=torch.int32)
    fid = (torch.arange(Rm) // (Rm // 2)).to(torch.int32)                                   # first half frame 0, second half frame 1
    feats0 = [0.3 * torch.randn(8, generator=gen) for _ in range(2)]
    Wm = {k: v.clone() for k, v in W.items()}
    train = list(steps.GEO_DECODER_PARAMS) + [n for n in steps.COLOR_DECODER_PARAMS if n in Wm]
    for n in train + [k for k in Wm if 'mlp_exposure' in k]:
        Wm[n].requires_grad_(True)
    geo_p, col_p = geo[rows.long()].clone().requires_grad_(True), col[rows.long()].clone().requires_grad_(True)
    fo = [f.clone().requires_grad_(True) for f in feats0]
    # Mapper.py:524-570: mlp_exposure belongs to color_decoder.parameters() (decoders_lr of the stage); only the CURRENT frame's
    # exposure feature (the last of the window) is an Adam parameter (lr 1e-3), the keyframe's is a constant
    opt = torch.optim.Adam([{'params': [Wm[n] for n in train] + [v for k, v in Wm.items() if 'mlp_exposure' in k], 'lr': 0.005},
                            {'params': [geo_p], 'lr': 0.005}, {'params': [col_p], 'lr': 0.005}, {'params': [fo[-1]], 'lr': 0.001}])
    om_losses = []
    mcfg = H.RenderCfg(rel_pos=False, exposure=True)
    for it in range(iters_m):
        opt.zero_grad()
        geo_t, col_t = geo.clone(), col.clone()
        geo_t[rows.long()], col_t[rows.long()] = geo_p, col_p
        ro, rd, gd, gc, _, _ = oracle_rays(c2w, depth_img, color_img, rnd_m[it])
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        out = H.render_batch(mcfg, ro[keep], rd[keep], gd[keep], pos, geo_t, col_t, Wm, 'color', color_sigmoid=False)
        color = out['color'].clone()
        f_keep = fid[keep].long()
        aff = torch.stack([H.exposure_affine(Wm, f) for f in fo])
        color = torch.sigmoid(torch.einsum('rc,rcd->rd', color, aff[:, :9].reshape(-1, 3, 3)[f_keep]) + aff[:, 9:][f_keep])
        m = (gd[keep] > 0) & out['valid_ray'] & (~torch.isnan(out['depth']))
        loss = torch.abs(gd[keep] - out['depth'])[m].sum() + 0.1 * torch.abs(gc[keep] - color)[m].sum()
        loss.backward()
        opt.step()
        om_losses.append(loss.item())
    dec2 = core.DecoderBlob(eng).pack(W)
    geo_d2, col_d2 = eng.f32(geo).clone(), eng.f32(col).clone()
    mlp2 = _exposure_module(W).to(eng.device)
    fk = [eng.f32(f).clone().requires_grad_(True) for f in feats0]
    mo = steps.MapOptimizer(eng, cfg, dec2, knn, pos_d, geo_d2, col_d2, rows.to(eng.device), Rm, lrs, w_color=0.1, exposure=(mlp2, fk))
    mo.begin_frame()
    frames = (eng.f32(depth_img).reshape(1, HH, WW).repeat(2, 1, 1), eng.f32(color_img).reshape(1, HH, WW, 3).repeat(2, 1, 1, 1),
              eng.f32(c2w).reshape(1, 4, 4).repeat(2, 1, 1), None)
    km = []
    if native:
        mlog = eng.zeros(iters_m, 4)
        mo.run(iters_m, 0, frames, rnd_m.to(eng.device), fid.to(eng.device), (0, HH, 0, WW), INTR, HH, WW, mlog)
        km = [float(x) for x in mlog[:, 0].cpu()]
    else:
        mo.native_loop = False
        for it in range(iters_m):
            out4 = mo.iterate('color', frames, rnd_m[it].to(eng.device), fid.to(eng.device), (0, HH, 0, WW), INTR, HH, WW)
            km.append(float(out4[0].cpu()))
    mo.finish()                                     # stacked exposure features -> the keyframes' tensors
    np.testing.assert_allclose(km, om_losses, rtol=5e-4)
    for a, b in zip(fk, fo):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), atol=2e-4)
    assert torch.equal(fk[0].detach().cpu(), feats0[0])                                     # the keyframe's feature is a constant
    assert float((fk[1].detach().cpu() - feats0[1]).abs().max()) > 1e-4                    # the current frame's moved
    np.testing.assert_allclose(mlp2[2].bias.detach().cpu().numpy(), Wm['color_decoder.mlp_exposure.linear2.bias'].detach().numpy(), atol=3e-4)
