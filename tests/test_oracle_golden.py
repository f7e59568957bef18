"""Pin the CPU oracle against golden vectors captured from the imported reference.

tests/golden/*.npz were produced by tools/gen_golden.py (which runs the
reference's own functions on CPU).  Tolerances are stated per test; fp32
throughout.
"""
import math

import numpy as np
import pytest
import torch

from oracle import hotpath as H
from util import load, tens, weights, CFG, CFG_NAMES, relerr

torch.set_num_threads(1)


def make_cfg(name, coef=0.1):
    c = CFG[name]
    return H.RenderCfg(S=5, near_surface=c['near_surface'], far_surface=c['far_surface'], near_end=0.3,
                       coef=coef, k=8, min_nn=2, radius_query=0.08, rel_pos=c['rel_pos'], exposure=c['exposure'])


def r2_of(g, name):
    if CFG[name]['dynamic']:
        return (torch.from_numpy(g['r_query']) ** 2).float()         # f64 square, then fp32
    return None


# ------------------------------------------------------------------ G1 composite
def test_g1_composite():
    g = load('g1_composite')
    raw, z = tens(g, 'raw', 'z')
    d, v, c, w = H.composite(raw[..., 3], raw[..., :3], z, float(g['coef']))
    for a, k in ((d, 'depth'), (v, 'var'), (c, 'rgb'), (w, 'w')):
        np.testing.assert_allclose(a.numpy(), g[k], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ G2 rays
@pytest.mark.parametrize('name', CFG_NAMES)
def test_g2_rays(name):
    g = load('g2_rays')
    i, j, c2w = tens(g, f'{name}_i', f'{name}_j', f'{name}_c2w')
    fx, fy, cx, cy = [float(x) for x in g[f'{name}_intr']]
    ro, rd = H.rays_from_uv(i, j, c2w, fx, fy, cx, cy)
    assert np.array_equal(ro.numpy(), g[f'{name}_rays_o'])
    assert np.array_equal(rd.numpy(), g[f'{name}_rays_d'])


def test_g2_image_rays():
    g = load('g2_rays')
    Hh, Ww, fx, fy, cx, cy, crop = g['img_params']
    ro, rd = H.image_rays(int(Hh), int(Ww), fx, fy, cx, cy, torch.from_numpy(g['tum_c2w']), crop_edge=int(crop))
    np.testing.assert_allclose(rd.numpy(), g['img_rays_d'].reshape(-1, 3), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ro.numpy(), g['img_rays_o'].reshape(-1, 3), rtol=0, atol=0)


# ------------------------------------------------------------------ G3 pose
def test_g3_pose():
    g = load('g3_pose')
    cams = torch.from_numpy(g['cams'])
    for n in range(cams.shape[0]):
        np.testing.assert_allclose(H.quat_to_c2w(cams[n]).numpy(), g['c2w'][n], rtol=1e-6, atol=1e-6)
        M = np.eye(4)
        M[:3] = g['c2w'][n]
        np.testing.assert_allclose(H.c2w_to_cam(M).numpy(), g['back'][n], rtol=1e-6, atol=1e-6)
    cam = cams[3].clone().requires_grad_(True)
    (H.quat_to_c2w(cam) * torch.from_numpy(g['grad_w'])).sum().backward()
    np.testing.assert_allclose(cam.grad.numpy(), g['grad_cam3'], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ embedding arithmetic contract
def test_embed_fma_order():
    """(2*pi*x) @ B with K=3 on torch-CPU == a0*b0 then two fmas in k order (bit-exact).

    This is the sequence the HIP kernels use for every Fourier embedding argument."""
    torch.manual_seed(0)
    p = torch.rand(5000, 3) * 6 - 3
    B = torch.randn(3, 93) * 25
    ref = ((H.TWO_PI * p) @ B).numpy()
    a = (H.TWO_PI * p).numpy().astype(np.float64)
    b = B.numpy().astype(np.float64)
    acc = (a[:, 0:1] * b[0:1]).astype(np.float32)
    acc = (a[:, 1:2] * b[1:2] + acc.astype(np.float64)).astype(np.float32)
    acc = (a[:, 2:3] * b[2:3] + acc.astype(np.float64)).astype(np.float32)
    assert (acc != ref).mean() < 1e-4       # double-rounding corner cases only


# ------------------------------------------------------------------ G4 kNN + interpolation
@pytest.mark.parametrize('name', CFG_NAMES)
def test_g4_knn_and_interp(name):
    g = load(f'g4_interp_{name}')
    W = weights(name)
    p, pos, geo, col = tens(g, 'p', 'pos', 'geo', 'col')
    if CFG[name]['dynamic']:
        r2 = (torch.from_numpy(g['r_pts']).reshape(-1) ** 2).float()
        r2_np = r2.numpy()
    else:
        r2 = torch.tensor(np.float32(0.08 ** 2))
        r2_np = np.float32(0.08 ** 2)
    d2, idx, cnt = H.knn_exact(pos.numpy(), p.numpy(), 8, r2_np)
    # against the reference stand-in's exact top-8: all in-radius entries must agree bit-for-bit
    D, I, nn = g['D'], g['I'], g['nn']
    rr = np.broadcast_to(np.asarray(r2_np).reshape(-1, 1), D.shape)
    inr = D <= rr
    assert np.array_equal(np.where(inr, I, -1), idx)
    assert np.array_equal(np.where(inr, D, np.float32(H.FLT_MAX)), d2)
    assert np.array_equal(nn, cnt)
    assert (cnt == 1).any() and (cnt == 8).any() and (cnt == 0).any()      # edge cases are present
    d2, idx, cnt = torch.from_numpy(d2), torch.from_numpy(idx), torch.from_numpy(cnt)
    ng, nc = tens(g, 'noise_geo', 'noise_col')
    for trk in (False, True):
        cg, has = H.interpolate(p, pos, geo, idx, d2, r2, cnt, 2, ng, trk)
        cc, _ = H.interpolate(p, pos, col, idx, d2, r2, cnt, 2, nc, trk,
                              relpos=H.relpos_params(W) if CFG[name]['rel_pos'] else None)
        assert np.array_equal(has.numpy(), g[f'has_trk{int(trk)}'])
        np.testing.assert_allclose(cg.numpy(), g[f'c_geo_trk{int(trk)}'], rtol=2e-5, atol=2e-7)
        np.testing.assert_allclose(cc.numpy(), g[f'c_col_trk{int(trk)}'], rtol=2e-5, atol=2e-6)


# ------------------------------------------------------------------ G5 decoders
@pytest.mark.parametrize('name', CFG_NAMES)
def test_g5_decoders(name):
    g = load(f'g5_mlp_{name}')
    W = weights(name)
    p, cg, cc = tens(g, 'p', 'c_geo', 'c_col')
    occ = H.geo_mlp(p, cg, W)
    np.testing.assert_allclose(occ.numpy(), g['occ'], rtol=1e-5, atol=1e-5)
    if CFG[name]['exposure']:
        aff = H.exposure_affine(W, torch.from_numpy(g['exposure_feat']))
        np.testing.assert_allclose(aff.numpy(), g['affine'], rtol=1e-6, atol=1e-7)
        rgb = H.color_mlp(p, cc, W, affine=aff)
        logits = H.color_mlp(p, cc, W, sigmoid=False)
        hm = g['has']        # the reference drew fresh noise for no-neighbour rows in this call
        np.testing.assert_allclose(logits.numpy()[hm], g['rgb_logits'][hm], rtol=1e-5, atol=1e-5)
    else:
        rgb = H.color_mlp(p, cc, W)
    np.testing.assert_allclose(rgb.numpy(), g['rgb'], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ G6 end-to-end render + grads
def _leafW(W):
    return {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in W.items()}


@pytest.mark.parametrize('stage', ('geometry', 'color'))
@pytest.mark.parametrize('name', CFG_NAMES)
def test_g6_mapper(name, stage):
    g = load(f'g6_render_{name}_map_{stage}')
    W = _leafW(weights(name))
    ro, rd, gd, gc, pos, geo, col = tens(g, 'rays_o', 'rays_d', 'gt_depth', 'gt_color', 'pos', 'geo', 'col')
    geo.requires_grad_(True)
    col.requires_grad_(True)
    cfg = make_cfg(name)
    expo = CFG[name]['exposure']
    out = H.render_batch(cfg, ro, rd, gd, pos, geo, col, W, stage, tracker=False, r2_ray=r2_of(g, name),
                         noise_geo=torch.from_numpy(g['noise_geo']),
                         noise_col=torch.from_numpy(g['noise_col']) if 'noise_col' in g else None,
                         color_sigmoid=not expo)
    assert np.array_equal(out['valid_ray'].numpy(), g['valid_ray'])
    np.testing.assert_allclose(out['depth'].detach().numpy(), g['depth'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['var'].detach().numpy(), g['var'], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(out['color'].detach().numpy(), g['color'], rtol=1e-4, atol=1e-5)
    color = out['color']
    if expo and stage == 'color':
        aff = H.exposure_affine(W, torch.from_numpy(g['exposure_feat']))
        color = torch.sigmoid(color @ aff[:9].reshape(3, 3) + aff[-3:])
    loss, geo_l, col_l, m = H.mapper_loss(out['depth'], color, out['valid_ray'], gd, gc, stage, float(g['w_color']))
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    loss.backward()
    assert relerr(geo.grad, g['grad_geo']) < 1e-4
    if 'grad_col' in g:
        assert relerr(col.grad, g['grad_col']) < 1e-4
    n_checked = 0
    for k, gv in g.items():
        if not k.startswith('gradW.'):
            continue
        mine = W[k[6:]].grad
        assert mine is not None, k
        assert relerr(mine, gv) < 2e-4, (k, relerr(mine, gv))
        n_checked += 1
    assert n_checked >= (1 if stage == 'geometry' else 20)


@pytest.mark.parametrize('name', CFG_NAMES)
def test_g6_tracker(name):
    g = load(f'g6_render_{name}_track')
    W = _leafW(weights(name))
    gd, gc, pos, geo, col, i, j = tens(g, 'gt_depth', 'gt_color', 'pos', 'geo', 'col', 'i', 'j')
    cam = torch.from_numpy(g['cam']).clone().requires_grad_(True)
    fx, fy, cx, cy = [float(x) for x in g['intr']]
    c2w = H.quat_to_c2w(cam)
    ro, rd = H.rays_from_uv(i, j, c2w, fx, fy, cx, cy)
    ro = ro.clone()
    ro.retain_grad()
    rd.retain_grad()
    np.testing.assert_allclose(rd.detach().numpy(), g['rays_d'], rtol=1e-6, atol=1e-7)
    cfg = make_cfg(name)
    aff = None
    if CFG[name]['exposure']:
        ef = torch.from_numpy(g['exposure_feat']).clone().requires_grad_(True)
        aff = H.exposure_affine(W, ef)
    out = H.render_batch(cfg, ro, rd, gd, pos, geo, col, W, 'color', tracker=True, r2_ray=r2_of(g, name),
                         noise_geo=torch.from_numpy(g['noise_geo']), noise_col=torch.from_numpy(g['noise_col']),
                         affine=aff)
    np.testing.assert_allclose(out['depth'].detach().numpy(), g['depth'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['var'].detach().numpy(), g['var'], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(out['color'].detach().numpy(), g['color'], rtol=1e-4, atol=1e-5)
    loss, geo_l, col_l, m = H.tracker_loss(out['depth'], out['var'], out['color'], gd, gc, float(g['w_color']))
    assert np.array_equal(m.numpy(), g['mask'])
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    loss.backward()
    assert relerr(rd.grad, g['grad_rays_d']) < 2e-4
    assert relerr(ro.grad, g['grad_rays_o']) < 2e-4
    assert relerr(cam.grad, g['grad_cam']) < 2e-4
    if CFG[name]['exposure']:
        assert relerr(ef.grad, g['grad_exposure_feat']) < 2e-4
        for k, gv in g.items():
            if k.startswith('gradW.'):
                assert relerr(W[k[6:]].grad, gv) < 2e-4, k


@pytest.mark.parametrize('name', CFG_NAMES)
def test_g6_img_zero_depth(name):
    """render_img-style batch: zero-depth rays sample linspace(near_end, far) and render depth 0."""
    g = load(f'g6_render_{name}_img')
    W = weights(name)
    ro, rd, gd, pos, geo, col = tens(g, 'rays_o', 'rays_d', 'gt_depth', 'pos', 'geo', 'col')
    cfg = make_cfg(name)
    aff = H.exposure_affine(W, torch.from_numpy(g['exposure_feat'])) if CFG[name]['exposure'] else None
    with torch.no_grad():
        out = H.render_batch(cfg, ro, rd, gd, pos, geo, col, W, 'color', r2_ray=r2_of(g, name),
                             noise_geo=torch.from_numpy(g['noise_geo']), noise_col=torch.from_numpy(g['noise_col']),
                             affine=aff)
    assert (gd == 0).any()
    assert np.array_equal(out['valid_ray'].numpy(), g['valid_ray'])
    np.testing.assert_allclose(out['depth'].numpy(), g['depth'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['var'].numpy(), g['var'], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(out['color'].numpy(), g['color'], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ G8 Adam
def test_g8_adam():
    g = load('g8_adam')
    P = [torch.from_numpy(g[k]).clone() for k in ('dec0', 'geo0', 'col0')]
    M = [torch.zeros_like(x) for x in P]
    V = [torch.zeros_like(x) for x in P]
    steps = [0, 0, 0]
    n_geo = int(g['n_geo_stage'])
    for it in range(20):
        stage_geo = it < n_geo
        lrs = (0.001, 0.03, 0.0) if stage_geo else (0.005, 0.005, 0.005)
        grads = [torch.from_numpy(g[k][it]) for k in ('gd', 'gg', 'gc')]
        for t in range(3):
            if t == 2 and stage_geo:
                continue                     # parameter without a gradient is skipped, its step count stays 0
            steps[t] += 1
            H.adam_step(P[t], grads[t], M[t], V[t], lrs[t], steps[t])
        for t, k in enumerate(('dec', 'geo', 'col')):
            np.testing.assert_allclose(P[t].numpy(), g[k][it], rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------ G9-G12 map maintenance (reference methods run on stand-ins)
@pytest.mark.parametrize('case', ('static', 'grad', 'dynamic', 'empty'))
def test_g9_add_points(case):
    """oracle.add_points vs NeuralPointCloud.add_neural_points (neural_point.py:1557-1631) with the radius rules of its
    find_neighbors_faiss call (step 'add': radius_add, radius_min for gradient points, per-ray dynamic radius)."""
    g = load('g9_add_points')
    ro, rd, cloud = tens(g, 'rays_o', 'rays_d', 'cloud')
    gd = torch.from_numpy(g[f'{case}_depth'])
    if case == 'dynamic':
        r2 = (torch.from_numpy(g['dynamic_radius']) ** 2).float().numpy()         # f64 square, then fp32 (the ABI's radius format)
    else:
        r2 = np.float32(float(g['radius_min' if case == 'grad' else 'radius_add']) ** 2)
    acc, pts = H.add_points(ro, rd, gd, cloud[:0] if case == 'empty' else cloud, r2, float(g['near_surface']), float(g['far_surface']), 3)
    assert acc.numel() == int(g[f'{case}_count']) and int(g[f'{case}_index_size']) == (0 if case == 'empty' else cloud.shape[0]) + pts.shape[0]
    assert np.array_equal(pts.numpy(), g[f'{case}_new_points'])                   # same expression, same order: bit for bit
    surf = ro[acc.long()] + rd[acc.long()] * gd[acc.long(), None]
    assert np.array_equal(surf.numpy(), g[f'{case}_input_pos'])
    np.testing.assert_allclose((torch.from_numpy(g['gt_color'])[acc.long()] * 255).numpy(), g[f'{case}_input_rgb'], rtol=0, atol=1e-4)


def test_g10_sample_near_pcl():
    g = load('g10_sample_near_pcl')
    z, invalid = H.sample_near_pcl(g['rays_o'], g['rays_d'], float(g['near']), float(g['far']), int(g['num']), g['cloud'], float(g['radius_query']))
    assert np.array_equal(invalid, g['invalid']) and 0 < invalid.sum() < invalid.size
    assert np.array_equal(z, g['z'])
    assert (np.abs(z[~invalid] - z[invalid][0]).max(axis=1) > 0.1).any()          # some rays really moved their samples to the cloud


def test_g11_filter_point_before_add():
    g = load('g11_filter_before_add')
    fx, fy, cx, cy = (float(x) for x in g['intr'])
    out = H.filter_point_before_add(*tens(g, 'rays_o', 'rays_d', 'gt_depth'), g['prev_c2w'], fx, fy, cx, cy, int(g['HW'][0]), int(g['HW'][1]))
    assert np.array_equal(out.numpy(), g['outside']) and 0.1 < g['outside'].mean() < 0.9


def test_g12_keyframe_overlap():
    """percent_inside of every keyframe as Mapper.keyframe_selection_overlap computes it (Mapper.py:250-270), on the rays the
    reference drew; the keyframes it selects are those with a non-zero overlap."""
    g = load('g12_keyframe_overlap')
    ro, rd, gd = tens(g, 'rays_o', 'rays_d', 'gt_depth')
    t = torch.linspace(0., 1., steps=int(g['N_samples']))
    dd = gd.reshape(-1, 1).repeat(1, int(g['N_samples']))
    z = (dd * 0.8) * (1. - t) + (dd + 0.5) * t
    pts = (ro[..., None, :] + rd[..., None, :] * z[..., :, None]).reshape(-1, 3).numpy()
    fx, fy, cx, cy = (float(x) for x in g['intr'])
    frac = H.keyframe_overlap_fractions(pts, g['est_c2ws'], fx, fy, cx, cy, int(g['HW'][0]), int(g['HW'][1]))
    np.testing.assert_allclose(frac, g['percent_inside'], rtol=0, atol=1e-12)
    assert sorted(np.nonzero(frac > 0)[0].tolist()) == g['selected'].tolist() and (frac == 0).any()
