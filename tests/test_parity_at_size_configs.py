"""HIP path vs the CPU oracle at the benchmarked sizes FOR WHAT MAKES THE TUM AND SCANNET CONFIGS DIFFERENT:
per-ray dynamic query radii from the colour-gradient pre-pass (lk_radius_maps; Tracker.py:243-268, Mapper.py:854-872), depth holes
in the training batches (rays the reference drops by boolean indexing stay in the static batch as 'absent'), ScanNet's surface
ratios 0.96 / 1.04 (scannet.yaml), the exposure affine - per keyframe on the rendered colour logits in the mapper
(Mapper.py:697-715), per sample inside the colour decoder in the tracker (decoder.py:534-540) - with gradients to mlp_exposure, the
exposure features and the feature rows, and half-precision feature tables on the 5 M-point cloud (BASELINE config 5).
10 000-ray mapping batches / 5 000-ray tracking batches over 100 000 points: several workgroups per compute unit, the launch geometry
of the benchmark.  Measured errors go to gpurun_out/parity_at_size.json beside those of tests/test_parity_at_size.py."""
import os

import numpy as np
import pytest
import torch

import atsize as A
from oracle import hotpath as H
from loopy_slam_amd import _ffi, core, optim, steps, synthetic as syn
from util import make_engine
from test_parity_at_size import _record, _check_grad, _gpu_scene, TOL_OUT, TOL_VAR
from test_steps_parity import _exposure_module

pytestmark = pytest.mark.gpu
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))

MODELS = {
    'tum': dict(near=0.98, far=1.02, exposure=False),
    'scannet': dict(near=0.96, far=1.04, exposure=True),
}


def _ocfg(m):
    return H.RenderCfg(S=5, near_surface=m['near'], far_surface=m['far'], near_end=0.3, coef=0.1, k=8, min_nn=2, radius_query=0.08,
                       rel_pos=False, exposure=m['exposure'])


def _batch(eng, R, frames, seed, window=None):
    """R random pixels spread over `frames` (2 % depth holes each) with the per-pixel dynamic query radius of their frame
    (lk_radius_maps on the frame's colour image), then the inside mask (Mapper.py:674-681) on the device.  Returns the CPU batch
    (kept = rays that are present for the losses) and the device tensors."""
    g = torch.Generator().manual_seed(seed * 7919 + R)
    H0, H1, W0, W1 = window if window is not None else (0, A.I['H'], 0, A.I['W'])
    i = torch.randint(W0, W1, (R,), generator=g).float()
    j = torch.randint(H0, H1, (R,), generator=g).float()
    fid = (torch.arange(R) * len(frames) // R).to(torch.int32)
    ro, rd, gd, gc, r2 = torch.empty(R, 3), torch.empty(R, 3), torch.empty(R), torch.empty(R, 3), torch.empty(R)
    c2ws = []
    for f, k in enumerate(frames):
        depth, color, c2w = syn.render_frame(k, device='cpu', holes=0.02)
        _, _, r2q = optim.radius_maps(eng, eng.f32(color), 0.15, 0.08, 0.02, 2.0)       # pointcloud.* of every config (Appendix B)
        m = fid == f
        o_, d_ = syn.pixel_rays(c2w, i[m], j[m])
        ro[m], rd[m] = o_, d_
        gd[m] = depth[j[m].long(), i[m].long()]
        gc[m] = color[j[m].long(), i[m].long()]
        r2[m] = r2q.cpu()[j[m].long(), i[m].long()]
        c2ws.append(c2w)
    dgd = eng.f32(gd)
    optim.inside_mask(eng, dgd, None, eng.empty(1), eng.empty(R, dtype=torch.int32), depth_filtered=dgd)
    gdf = dgd.cpu()
    thr = H.inside_threshold(gd[gd > 0])
    keep = (gd > 0) & (gd <= thr)
    assert torch.equal(gdf > 0, keep) and torch.equal(gdf[keep], gd[keep])          # the device mask is the reference's
    b = dict(i=i, j=j, fid=fid, rays_o=ro, rays_d=rd, gt_depth=gdf, gt_color=gc, r2=r2, keep=keep, c2w=c2ws)
    dev = dict(ro=eng.f32(ro), rd=eng.f32(rd), gd=dgd, gc=eng.f32(gc), r2=eng.f32(r2), fid=fid.to(eng.device))
    return b, dev


def _knn_of_kept(st, b, pos, m, case):
    """neighbour lists / counts / sample depths of the PRESENT rays against the contract with the per-ray radius: bit-exact."""
    keep = b['keep']
    z, _ = H.sample_z(b['gt_depth'][keep], m['near'], m['far'], 0.3, 5)
    assert np.array_equal(st.z.cpu()[keep].numpy(), z.numpy())
    p = H.sample_points(b['rays_o'][keep], b['rays_d'][keep], z)
    r2p = b['r2'][keep].reshape(-1, 1).repeat(1, 5).reshape(-1).numpy()
    ks = keep.reshape(-1, 1).repeat(1, 5).reshape(-1)
    got = st.nbr_idx.cpu()[ks].numpy()
    d2, idx, cnt, n_re = A.contract_knn(pos, p, r2p, got_idx=got)
    assert np.array_equal(got, idx), f'{int((got != idx).any(1).sum())} neighbour lists differ from the contract'
    assert np.array_equal(st.nbr_count.cpu()[ks].numpy(), cnt)
    _record(case, knn_rows=int(idx.shape[0]), knn_rows_rechecked_brute_force=n_re, absent_rays=int((~keep).sum()),
            r_query_min=float(np.sqrt(r2p.min())), r_query_max=float(np.sqrt(r2p.max())))
    return d2, idx, cnt


def _fwd_check(st, o, keep, case, logits=False):
    d, c, v = st.depth.cpu()[keep], st.color.cpu()[keep], st.var.cpu()[keep]
    _record(case, depth_rel=A.errs(d, o['depth'].detach())[0], color_rel=A.errs(c, o['color'].detach())[0], var_rel=A.errs(v, o['var'].detach())[0])
    assert np.array_equal(st.valid_ray.cpu()[keep].numpy().astype(bool), o['valid_ray'].numpy())
    np.testing.assert_allclose(d.numpy(), o['depth'].detach().numpy(), rtol=TOL_OUT, atol=1e-6)
    np.testing.assert_allclose(c.numpy(), o['color'].detach().numpy(), rtol=TOL_OUT, atol=2e-5 if not logits else 1e-4)
    np.testing.assert_allclose(v.numpy(), o['var'].detach().numpy(), rtol=TOL_VAR, atol=1e-9)
    assert float(st.depth.cpu()[~keep].abs().max()) == 0.0                          # absent rays render depth 0 (Renderer.py:197-198)


@pytest.mark.parametrize('model,stage', (('tum', 'geometry'), ('tum', 'color'), ('scannet', 'color')))
def test_mapper_iteration_dynamic_radius_holes_exposure(model, stage):
    """One mapping iteration at 10 000 rays over four keyframes: dynamic radii + holes (TUM, ScanNet), ScanNet ratios and the
    per-keyframe exposure affine with lk_loss_mapper_exposure - forward, loss and every gradient against oracle autograd."""
    m = MODELS[model]
    R, F = 10_000, 4
    case = f'map-{stage}-{model}-R{R}-dynamic-holes' + ('-exposure' if m['exposure'] else '')
    eng = make_engine('hip')
    pos, geo, col = A.scene(100_000)
    W = syn.default_weights(rel_pos=False, exposure=m['exposure'])
    dpos, dgeo, dcol = eng.f32(pos), eng.f32(geo), eng.f32(col)
    knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(dpos)
    dec = core.DecoderBlob(eng).pack(W)
    b, dv = _batch(eng, R, frames=(3, 7, 12, 18), seed=4)
    keep = b['keep']
    cfg = core.RenderCfg(rel_pos=False, near_surface=m['near'], far_surface=m['far'], exposure=m['exposure'])
    st = core.RenderState(eng, R, cfg.S, need_act=True)
    d_depth, d_color, out4 = eng.empty(R), eng.empty(R, 3), eng.zeros(4)
    xs = None
    if m['exposure']:
        gen = torch.Generator().manual_seed(8)
        feats0 = [0.3 * torch.randn(8, generator=gen) for _ in range(F)]
        mlp = _exposure_module(W).to(eng.device)
        xs = steps.ExposureState(eng, mlp, [eng.f32(f).clone().requires_grad_(True) for f in feats0])
        xs.forward()
        core.render_forward(eng, cfg, st, dv['ro'], dv['rd'], dv['gd'], knn, dpos, dgeo, dcol, dec, stage, r2_ray=dv['r2'], save_act=True,
                            extra_flags=_ffi.FLAG_ZERO_ABSENT, color_logits=True)
        eng.lib.check(eng.lib.dll.lk_loss_mapper_exposure(R, _ffi.ptr(st.depth), _ffi.ptr(st.color), _ffi.ptr(st.valid_ray), _ffi.ptr(dv['gd']),
                                                          _ffi.ptr(dv['gc']), _ffi.ptr(dv['fid']), _ffi.ptr(xs.aff), F, _ffi.C.c_float(0.1),
                                                          _ffi.ptr(d_depth), _ffi.ptr(d_color), _ffi.ptr(out4), _ffi.ptr(xs.g_aff), eng.stream),
                      'lk_loss_mapper_exposure')
    else:
        core.render_forward(eng, cfg, st, dv['ro'], dv['rd'], dv['gd'], knn, dpos, dgeo, dcol, dec, stage, r2_ray=dv['r2'], save_act=True,
                            extra_flags=_ffi.FLAG_ZERO_ABSENT, mapper_loss=(dv['gc'], 0.1, d_depth, d_color, out4))
    torch.cuda.synchronize()
    kn = _knn_of_kept(st, b, pos, m, case)
    ro_, rd_, gd_, gc_, r2k, fk = b['rays_o'][keep], b['rays_d'][keep], b['gt_depth'][keep], b['gt_color'][keep], b['r2'][keep], b['fid'][keep].long()
    gd, gc = gd_, gc_

    def oracle(exclude=None, grads=True, f64=False):
        c = A.to64 if f64 else (lambda x: x)             # f64: the float64 referee (inside atsize.ref64)
        names = [k for k in W if k != 'color_decoder.embedder._B']
        Wr = {k: c(v).clone().requires_grad_(grads and k in names) for k, v in W.items()}
        geo_r, col_r = c(geo).clone().requires_grad_(grads), c(col).clone().requires_grad_(grads)
        fo = [c(f).clone().requires_grad_(grads) for f in feats0] if m['exposure'] else None
        ro, rd, gd, gc = c(ro_), c(rd_), c(gd_), c(gc_)
        o = H.render_batch(_ocfg(m), ro, rd, gd, c(pos), geo_r, col_r, Wr, stage, r2_ray=r2k, knn=c(tuple(kn)), color_sigmoid=not m['exposure'])
        color = o['color']
        if m['exposure']:       # Mapper.py:697-715: the keyframe's affine on the composited logits, then the sigmoid
            aff = torch.stack([H.exposure_affine(Wr, f) for f in fo])
            color = torch.sigmoid(torch.einsum('rc,rcd->rd', o['color'], aff[:, :9].reshape(-1, 3, 3)[fk]) + aff[:, 9:][fk])
        valid = o['valid_ray'] if exclude is None else (o['valid_ray'] & ~exclude)
        loss = H.mapper_loss(o['depth'], color, valid, gd, gc, stage, 0.1)
        res = dict(out=o, loss=H.mapper_loss(o['depth'], color, o['valid_ray'], gd, gc, stage, 0.1))
        if grads:
            loss[0].backward()
            res.update(g_geo=geo_r.grad, g_col=col_r.grad, gW={k: Wr[k].grad for k in names if Wr[k].grad is not None},
                       g_feats=[f.grad for f in fo] if fo else None)
        return res
    with torch.no_grad():
        r0 = oracle(grads=False)
    _fwd_check(st, r0['out'], keep, case, logits=m['exposure'])
    loss, lgeo, lcol, msk = r0['loss']
    o4 = out4.cpu().numpy()
    _record(case, loss_rel=abs(o4[0] - float(loss)) / abs(float(loss)), masked=int(msk.sum()))
    assert abs(o4[0] - float(loss)) <= TOL_OUT * abs(float(loss)) and int(o4[3]) == int(msk.sum())
    assert abs(o4[1] - float(lgeo)) <= TOL_OUT * abs(float(lgeo))
    if stage == 'color':
        assert abs(o4[2] - float(lcol)) <= TOL_OUT * abs(float(lcol))
    # rays on a branch point of the graph: zero loss gradient on both sides
    bk = dict(gt_depth=gd)
    bp, margin = A.branch_point_rays(r0['out'], bk, pos, geo, W, r2=r2k.reshape(-1, 1).repeat(1, 5).reshape(-1))
    _record(case, branch_point_rays=int(bp.sum()), relu_margin_min=margin)
    assert int(bp.sum()) <= 120
    bp_full = torch.zeros(R, dtype=torch.bool)
    bp_full[torch.nonzero(keep).reshape(-1)[bp]] = True
    if m['exposure'] and int(bp.sum()):
        # the exposure loss kernel already summed d loss / d affine over every ray: re-evaluate it without the branch-point rays
        gdx = dv['gd'].clone()
        gdx[bp_full.to(eng.device)] = 0.0
        eng.lib.check(eng.lib.dll.lk_loss_mapper_exposure(R, _ffi.ptr(st.depth), _ffi.ptr(st.color), _ffi.ptr(st.valid_ray), _ffi.ptr(gdx),
                                                          _ffi.ptr(dv['gc']), _ffi.ptr(dv['fid']), _ffi.ptr(xs.aff), F, _ffi.C.c_float(0.1),
                                                          _ffi.ptr(d_depth), _ffi.ptr(d_color), _ffi.ptr(eng.zeros(4)), _ffi.ptr(xs.g_aff), eng.stream),
                      'lk_loss_mapper_exposure')
    elif int(bp.sum()):
        d_depth[bp_full.to(eng.device)] = 0.0
        d_color[bp_full.to(eng.device)] = 0.0
    gs = core.GradState(eng, pos.shape[0], R, dec.n, feats=True, weights=True)
    core.render_backward(eng, st, gs, d_depth, d_color)
    if xs is not None:
        xs.backward(xs.g_aff)
    torch.cuda.synchronize()
    r = oracle(exclude=bp)
    with A.ref64():
        q = oracle(exclude=bp, f64=True)
    _check_grad('geo_feats', gs.g_geo.cpu(), r['g_geo'], case, ref64=q['g_geo'])
    if stage == 'color':
        _check_grad('col_feats', gs.g_col.cpu(), r['g_col'], case, ref64=q['g_col'])
    gW = dec.unpack(gs.g_weights)
    n = 0
    for name, ref in r['gW'].items():
        if (name.startswith('geo_decoder.') and name != 'geo_decoder.embedder._B') or 'mlp_exposure' in name:
            continue
        if name not in gW or (stage == 'geometry' and not name.startswith('geo_decoder.')):
            continue
        _check_grad(name, gW[name].reshape(ref.shape), ref, case, ref64=q['gW'][name])
        n += 1
    assert n >= (1 if stage == 'geometry' else 22)
    if xs is not None:
        g = xs.g.cpu()
        X = 'color_decoder.mlp_exposure.'
        _check_grad('mlp_exposure.linear1.weight', g[0:1024].reshape(128, 8), r['gW'][X + 'linear1.weight'], case, ref64=q['gW'][X + 'linear1.weight'])
        _check_grad('mlp_exposure.linear1.bias', g[1024:1152], r['gW'][X + 'linear1.bias'], case, ref64=q['gW'][X + 'linear1.bias'])
        _check_grad('mlp_exposure.linear2.weight', g[1152:2688].reshape(12, 128), r['gW'][X + 'linear2.weight'], case, ref64=q['gW'][X + 'linear2.weight'])
        _check_grad('mlp_exposure.linear2.bias', g[2688:2700], r['gW'][X + 'linear2.bias'], case, ref64=q['gW'][X + 'linear2.bias'])
        _check_grad('exposure_feats', g[2700:2700 + 8 * F].reshape(F, 8), torch.stack(r['g_feats']), case, ref64=torch.stack(q['g_feats']))


@pytest.mark.parametrize('model', ('tum', 'scannet'))
def test_tracker_iteration_dynamic_radius_holes_exposure(model):
    """One tracking iteration at 5 000 rays (Tracker.py:142-195, 329-344): dynamic radii, holes, ScanNet ratios and the frame's
    exposure affine inside the colour decoder - loss, pose gradient and the exposure gradients against the oracle."""
    m = MODELS[model]
    R = 5_000
    case = f'track-{model}-R{R}-dynamic-holes' + ('-exposure' if m['exposure'] else '')
    eng = make_engine('hip')
    pos, geo, col = A.scene(100_000)
    W = syn.default_weights(rel_pos=False, exposure=m['exposure'])
    dpos, dgeo, dcol = eng.f32(pos), eng.f32(geo), eng.f32(col)
    knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(dpos)
    dec = core.DecoderBlob(eng).pack(W)
    b, dv = _batch(eng, R, frames=(5,), seed=6, window=(20, A.I['H'] - 20, 20, A.I['W'] - 20))
    keep = b['keep']
    cam = H.c2w_to_cam(b['c2w'][0])
    cfg = core.RenderCfg(rel_pos=False, near_surface=m['near'], far_surface=m['far'], exposure=m['exposure'])
    st = core.RenderState(eng, R, cfg.S, need_act=True)
    dcam, pi, pj = eng.f32(cam), eng.f32(b['i']), eng.f32(b['j'])
    ro, rd = eng.empty(R, 3), eng.empty(R, 3)
    optim.rays_from_pose(eng, dcam, pi, pj, A.INTR, ro, rd)
    xs, aff = None, None
    if m['exposure']:
        feat0 = 0.3 * torch.randn(8, generator=torch.Generator().manual_seed(9))
        xs = steps.ExposureState(eng, _exposure_module(W).to(eng.device), eng.f32(feat0).clone().requires_grad_(True))
        aff = xs.forward()[0]
    core.render_forward(eng, cfg, st, ro, rd, dv['gd'], knn, dpos, dgeo, dcol, dec, 'color', tracker=True, r2_ray=dv['r2'], save_act=True,
                        affine=aff, extra_flags=_ffi.FLAG_ZERO_ABSENT)
    d_depth, d_color, out4 = eng.empty(R), eng.empty(R, 3), eng.zeros(4)
    optim.loss_tracker(eng, st, dv['gd'], dv['gc'], 0.5, True, d_depth, d_color, out4, eng.empty(R + 8))
    torch.cuda.synchronize()
    bo = dict(b)
    bo['rays_o'], bo['rays_d'] = ro.cpu(), rd.cpu()
    kn = _knn_of_kept(st, bo, pos, m, case)
    ik, jk, gd_, gc_, r2k = b['i'][keep], b['j'][keep], b['gt_depth'][keep], b['gt_color'][keep], b['r2'][keep]
    gd, gc = gd_, gc_

    kro, krd = ro.cpu()[keep], rd.cpu()[keep]            # every evaluation AT THE KERNEL'S RAYS (atsize.oracle_tracker: rays_value)

    def oracle(exclude=None, grads=True, f64=False, var32=None):
        c = A.to64 if f64 else (lambda x: x)             # f64: the float64 referee (inside atsize.ref64), var32: atsize.oracle_tracker64
        cam_r = c(cam).clone().requires_grad_(grads)
        Wr = {k: (c(v).clone().requires_grad_(grads) if 'mlp_exposure' in k else c(v)) for k, v in W.items()}
        fo = c(feat0).clone().requires_grad_(grads) if m['exposure'] else None
        gd, gc = c(gd_), c(gc_)
        ro_o, rd_o = H.rays_from_uv(c(ik), c(jk), H.quat_to_c2w(cam_r), *A.INTR)
        ro_o, rd_o = c(kro) + (ro_o - ro_o.detach()), c(krd) + (rd_o - rd_o.detach())
        o = H.render_batch(_ocfg(m), ro_o, rd_o, gd, c(pos), c(geo), c(col), Wr, 'color', tracker=True, r2_ray=r2k, knn=c(tuple(kn)),
                           affine=H.exposure_affine(Wr, fo) if fo is not None else None)
        var = o['var'] if var32 is None else c(var32)
        loss = H.tracker_loss(o['depth'], var, o['color'], gd, gc, 0.5)
        res = dict(out=o, loss=loss)
        if grads:
            mk = loss[3] if exclude is None else (loss[3] & ~exclude)
            tmp = torch.abs(gd - o['depth']) / torch.sqrt(var.detach() + 1e-10)
            (torch.clamp(tmp, min=0.0, max=1e3)[mk].sum() + 0.5 * torch.abs(gc - o['color'])[mk].sum()).backward()
            res.update(g_cam=cam_r.grad, gW={k: v.grad for k, v in Wr.items() if 'mlp_exposure' in k}, g_feat=fo.grad if fo is not None else None)
        return res
    with torch.no_grad():
        r0 = oracle(grads=False)
    _fwd_check(st, r0['out'], keep, case)
    loss, lgeo, lcol, msk = r0['loss']
    o4 = out4.cpu().numpy()
    _record(case, loss_rel=abs(o4[0] - float(loss)) / abs(float(loss)), masked=int(msk.sum()))
    assert int(o4[3]) == int(msk.sum()) and abs(o4[0] - float(loss)) <= TOL_OUT * abs(float(loss))
    r2p = r2k.reshape(-1, 1).repeat(1, 5).reshape(-1)
    bp, margin = A.branch_point_rays(r0['out'], dict(gt_depth=gd, gt_color=gc), pos, geo, W, tracker_loss=True, r2=r2p)
    _record(case, branch_point_rays=int(bp.sum()), relu_margin_min=margin)
    assert int(bp.sum()) <= 120
    if int(bp.sum()):
        bp_full = torch.zeros(R, dtype=torch.bool)
        bp_full[torch.nonzero(keep).reshape(-1)[bp]] = True
        d_depth[bp_full.to(eng.device)] = 0.0
        d_color[bp_full.to(eng.device)] = 0.0
    gs = core.GradState(eng, pos.shape[0], R, dec.n, feats=False, weights=False, rays=True, affine=m['exposure'])
    core.render_backward(eng, st, gs, d_depth, d_color)
    g_cam = eng.zeros(7)
    optim.pose_bwd(eng, dcam, pi, pj, A.INTR, gs.g_rays_o, gs.g_rays_d, g_cam)
    if xs is not None:
        xs.backward(gs.g_affine)
    torch.cuda.synchronize()
    r = oracle(exclude=bp)
    with A.ref64():
        q = oracle(exclude=bp, f64=True, var32=r['out']['var'].detach())
    _check_grad('cam', g_cam.cpu(), r['g_cam'], case, ref64=q['g_cam'])
    assert float(gs.g_rays_d.cpu()[~keep].abs().max()) == 0.0                           # absent rays carry no gradient
    if xs is not None:
        g = xs.g.cpu()
        X = 'color_decoder.mlp_exposure.'
        _check_grad('mlp_exposure.linear1.weight', g[0:1024].reshape(128, 8), r['gW'][X + 'linear1.weight'], case, ref64=q['gW'][X + 'linear1.weight'])
        _check_grad('mlp_exposure.linear2.weight', g[1152:2688].reshape(12, 128), r['gW'][X + 'linear2.weight'], case, ref64=q['gW'][X + 'linear2.weight'])
        _check_grad('mlp_exposure.linear2.bias', g[2688:2700], r['gW'][X + 'linear2.bias'], case, ref64=q['gW'][X + 'linear2.bias'])
        _check_grad('exposure_feat', g[2700:2708], r['g_feat'], case, ref64=q['g_feat'])


def test_forward_with_half_tables_at_5m_points():
    """BASELINE config 5 ('fp16 features'): the 5 M-point cloud with IEEE-half feature tables (LK_FLAG_FEATS_F16), 12 000 rays with
    dynamic radii and holes, against the oracle on the same half-rounded tables."""
    case = 'fwd-N5000000-scannet-f16'
    m = MODELS['scannet']
    eng = make_engine('hip')
    N, R = 5_000_000, 12_000
    pos, geo, col = A.scene(N)
    W = syn.default_weights(rel_pos=False)
    geo_h, col_h = geo.half(), col.half()
    dpos = eng.f32(pos)
    knn = core.KnnIndex(eng, capacity=N); knn.build(dpos)
    dec = core.DecoderBlob(eng).pack(W)
    b, dv = _batch(eng, R, frames=(11, 40), seed=5)
    keep = b['keep']
    cfg = core.RenderCfg(rel_pos=False, near_surface=m['near'], far_surface=m['far'])
    st = core.RenderState(eng, R, cfg.S)
    core.render_forward(eng, cfg, st, dv['ro'], dv['rd'], dv['gd'], knn, dpos, geo_h.to(eng.device), col_h.to(eng.device), dec, 'color',
                        r2_ray=dv['r2'], extra_flags=_ffi.FLAG_ZERO_ABSENT)
    assert st.desc.flags & _ffi.FLAG_FEATS_F16
    torch.cuda.synchronize()
    kn = _knn_of_kept(st, b, pos, m, case)
    with torch.no_grad():
        o = H.render_batch(_ocfg(dict(m, exposure=False)), b['rays_o'][keep], b['rays_d'][keep], b['gt_depth'][keep], pos, geo_h.float(), col_h.float(),
                           W, 'color', r2_ray=b['r2'][keep], knn=kn)
    _fwd_check(st, o, keep, case)


def test_scannet_loops_fp16_pieces_with_device_scale_match_bf16_statement_path():
    """Exposure encoding inside the native loops runs the colour decoder's backward and the weight-gradient reductions on pre-scaled fp16
    pieces although the loss gradient passes through the LEARNED affines: the exposure step keeps a power of two on the device
    (lk_exposure_desc::bwd_scale) that the kernels apply on top of their 2^10.  At 10 000 rays over four keyframes: three colour
    iterations of lk_map_frame against the per-statement path (bf16 pieces, no scale) from the same start - losses, the exposure
    feature and the decoder after the steps - with affines scaled to 0.05x and 30x of their start so that the scale cell matters."""
    m = MODELS['scannet']
    R, F, iters = 10_000, 4, 3
    eng = make_engine('hip')
    pos, geo, col = A.scene(100_000)
    W0 = syn.default_weights(rel_pos=False, exposure=True)
    lrs = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
    dpos = eng.f32(pos)
    knn = core.KnnIndex(eng, capacity=pos.shape[0]); knn.build(dpos)
    frames_cpu = [syn.render_frame(k, device='cpu', holes=0.02) for k in (3, 7, 12, 18)]
    stack = (eng.f32(torch.stack([f[0] for f in frames_cpu])), eng.f32(torch.stack([f[1] for f in frames_cpu])),
             eng.f32(torch.stack([f[2] for f in frames_cpu])),
             torch.stack([optim.radius_maps(eng, eng.f32(f[1]), 0.15, 0.08, 0.02, 2.0)[2] for f in frames_cpu]).contiguous())
    gen = torch.Generator().manual_seed(77)
    rnd = torch.randint(0, A.I['H'] * A.I['W'], (iters, R), generator=gen, dtype=torch.int32).to(eng.device)
    fid = (torch.arange(R) * F // R).to(torch.int32).to(eng.device)
    rows = torch.arange(0, pos.shape[0], 2, dtype=torch.int32).to(eng.device)
    cfg = core.RenderCfg(rel_pos=False, near_surface=m['near'], far_surface=m['far'], exposure=True)
    feats0 = [0.3 * torch.randn(8, generator=gen) for _ in range(F)]
    for gain in (0.05, 30.0):
        W = {k: v.clone() for k, v in W0.items()}
        W['color_decoder.mlp_exposure.linear2.bias'] = W['color_decoder.mlp_exposure.linear2.bias'] * gain
        W['color_decoder.mlp_exposure.linear2.weight'] = W['color_decoder.mlp_exposure.linear2.weight'] * gain
        res = {}
        for native in (True, False):
            dec = core.DecoderBlob(eng).pack(W)
            dgeo, dcol = eng.f32(geo).clone(), eng.f32(col).clone()
            mlp = _exposure_module(W).to(eng.device)
            fk = [eng.f32(f).clone().requires_grad_(True) for f in feats0]
            mo = steps.MapOptimizer(eng, cfg, dec, knn, dpos, dgeo, dcol, rows, R, lrs, w_color=0.1, dynamic_radius=True, exposure=(mlp, fk))
            mo.native_loop = native
            mo.begin_frame()
            log = eng.zeros(iters, 4)
            mo.run(iters, 0, stack, rnd, fid, (0, A.I['H'], 0, A.I['W']), A.INTR, A.I['H'], A.I['W'], log)
            mo.finish()
            torch.cuda.synchronize()
            if native:
                scale = float(mo.exposure.bwd_scale.cpu())
                assert scale != 1.0 and abs(np.log2(scale) - round(np.log2(scale))) < 1e-6, scale        # a power of two that moved
            res[native] = (log[:, 0].cpu().numpy().copy(), fk[-1].detach().cpu().clone(), dec.blob.cpu().clone(), mlp[2].bias.detach().cpu().clone())
        ln, ls = res[True][0], res[False][0]
        _record(f'scannet-loops-fp16-scale-gain{gain}', loss_rel=float(np.abs(ln - ls).max() / np.abs(ls).max()), bwd_scale=scale)
        np.testing.assert_allclose(ln, ls, rtol=2e-5)
        np.testing.assert_allclose(res[True][1].numpy(), res[False][1].numpy(), atol=2e-5)                  # the current frame's exposure feature
        np.testing.assert_allclose(res[True][3].numpy(), res[False][3].numpy(), atol=2e-4 * max(1.0, gain))
        d = (res[True][2] - res[False][2]).abs()
        assert float(torch.quantile(d, 0.999)) < 2e-5 and float(d.max()) < 2 * 0.005 * iters + 1e-6        # Adam's sign-like first steps bound the tail
