"""Fused loss kernels, multi-tensor Adam, pose<->rays, inside-mask and compaction against the
oracle / golden vectors (both back-ends, see test_forward_parity.py)."""
import numpy as np
import pytest
import torch

from oracle import hotpath as H
from loopy_slam_amd import core, optim
from util import load, tens, make_engine, backends, relerr

torch.set_num_threads(1)


class _St:
    pass


def _fake_state(eng, depth, var, color, valid):
    st = _St()
    st.depth, st.var, st.color = eng.f32(depth), eng.f32(var), eng.f32(color)
    st.valid_ray = valid.to(torch.uint8).to(eng.device).contiguous()
    return st


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('stage', ('geometry', 'color'))
def test_loss_mapper(backend, stage):
    eng = make_engine(backend)
    g = load(f'g6_render_replica_map_{stage}')
    depth, var, color, gd, gc = tens(g, 'depth', 'var', 'color', 'gt_depth', 'gt_color')
    depth = depth.clone()
    depth[3] = float('nan')                       # NaN depth is masked (Mapper.py:692)
    valid = torch.from_numpy(g['valid_ray'])
    st = _fake_state(eng, depth, var, color, valid)
    R = depth.shape[0]
    dd, dc, out = eng.empty(R), eng.empty(R, 3), eng.empty(4)
    optim.loss_mapper(eng, st, eng.f32(gd), eng.f32(gc), float(g['w_color']), stage == 'color', dd, dc, out)
    dl = depth.clone().requires_grad_(True)
    cl = color.clone().requires_grad_(True)
    loss, geo, col, m = H.mapper_loss(dl, cl, valid, gd, gc, stage, float(g['w_color']))
    loss.backward()
    o = out.cpu().numpy()
    assert abs(o[0] - loss.item()) <= 1e-5 * abs(loss.item()) and abs(o[1] - geo.item()) <= 1e-5 * abs(geo.item())
    assert o[3] == int(m.sum())
    np.testing.assert_allclose(dd.cpu().numpy(), torch.nan_to_num(dl.grad).numpy(), atol=0)
    if stage == 'color':
        np.testing.assert_allclose(dc.cpu().numpy(), cl.grad.numpy(), atol=1e-7)


@pytest.mark.parametrize('backend', backends())
def test_loss_tracker(backend):
    eng = make_engine(backend)
    g = load('g6_render_tum_track')
    depth, var, color, gd, gc = tens(g, 'depth', 'var', 'color', 'gt_depth', 'gt_color')
    st = _fake_state(eng, depth, var, color, torch.ones(depth.shape[0], dtype=torch.bool))
    R = depth.shape[0]
    dd, dc, out, scr = eng.empty(R), eng.empty(R, 3), eng.empty(4), eng.empty(R + 8)
    optim.loss_tracker(eng, st, eng.f32(gd), eng.f32(gc), float(g['w_color']), True, dd, dc, out, scr)
    dl = depth.clone().requires_grad_(True)
    cl = color.clone().requires_grad_(True)
    loss, geo, col, m = H.tracker_loss(dl, var, cl, gd, gc, float(g['w_color']))
    loss.backward()
    o = out.cpu().numpy()
    assert abs(o[0] - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    assert o[3] == int(m.sum())
    np.testing.assert_allclose(dd.cpu().numpy(), dl.grad.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dc.cpu().numpy(), cl.grad.numpy(), atol=1e-7)


@pytest.mark.parametrize('backend', backends())
def test_losses_large_batch(backend):
    """Above 16 384 rays the losses take the multi-workgroup (atomic) kernels: synthetic 20 000-ray batch, absent rays
    (gt_depth 0), invalid rays and a NaN depth, against the oracle."""
    eng = make_engine(backend)
    gen = torch.Generator().manual_seed(11)
    R = 20000
    gd = torch.rand(R, generator=gen) * 3 + 0.5
    gd[torch.rand(R, generator=gen) < 0.05] = 0.0
    depth = gd + 0.05 * torch.randn(R, generator=gen)
    depth[7] = float('nan')
    var = torch.rand(R, generator=gen) * 0.01 + 1e-4
    color, gc = torch.rand(R, 3, generator=gen), torch.rand(R, 3, generator=gen)
    valid = torch.rand(R, generator=gen) < 0.9
    st = _fake_state(eng, depth, var, color, valid)
    dd, dc, out, scr = eng.empty(R), eng.empty(R, 3), eng.empty(4), eng.empty(R + 8)
    optim.loss_mapper(eng, st, eng.f32(gd), eng.f32(gc), 0.1, True, dd, dc, out)
    dl, cl = depth.clone().requires_grad_(True), color.clone().requires_grad_(True)
    loss, geo, col, m = H.mapper_loss(dl, cl, valid, gd, gc, 'color', 0.1)
    loss.backward()
    o = out.cpu().numpy()
    assert abs(o[0] - loss.item()) <= 2e-5 * abs(loss.item()) and o[3] == int(m.sum())
    np.testing.assert_allclose(dd.cpu().numpy(), torch.nan_to_num(dl.grad).numpy(), atol=0)
    np.testing.assert_allclose(dc.cpu().numpy(), cl.grad.numpy(), atol=1e-7)
    # tracker: the reference never sees absent rays (filtered before rendering) -> compare on the present ones
    keep = gd > 0
    depth2 = torch.nan_to_num(depth, nan=1.0)
    st2 = _fake_state(eng, depth2, var, color, torch.ones(R, dtype=torch.bool))
    optim.loss_tracker(eng, st2, eng.f32(gd), eng.f32(gc), 0.5, True, dd, dc, out, scr)
    dl, cl = depth2[keep].clone().requires_grad_(True), color[keep].clone().requires_grad_(True)
    loss, geo, col, m = H.tracker_loss(dl, var[keep], cl, gd[keep], gc[keep], 0.5)
    loss.backward()
    o = out.cpu().numpy()
    assert abs(o[0] - loss.item()) <= 2e-5 * abs(loss.item()) and o[3] == int(m.sum())
    np.testing.assert_allclose(dd.cpu().numpy()[keep.numpy()], dl.grad.numpy(), rtol=1e-6, atol=1e-7)
    assert float(dd.cpu()[~keep].abs().max()) == 0.0


@pytest.mark.parametrize('R', (1, 2, 777, 20000))
@pytest.mark.parametrize('backend', backends())
def test_loss_tracker_median_mask(backend, R):
    """tracking.handle_dynamic: False (Tracker.py:177-179): mask = |gt - depth| < 10 * median(|gt - depth|) (torch.median: the lower
    middle value), loss terms unchanged.  One-workgroup form (R <= 16 384) and the three-launch form; absent rays (gt 0) are not part
    of the reference's batch; outliers that the mask must reject; ties at the median; a NaN residual empties the mask."""
    eng = make_engine(backend)
    gen = torch.Generator().manual_seed(100 + R)
    gd = torch.rand(R, generator=gen) * 3 + 0.5
    if R > 2:
        gd[torch.rand(R, generator=gen) < 0.05] = 0.0
    depth = gd + 0.01 * torch.randn(R, generator=gen)
    out_l = torch.rand(R, generator=gen) < 0.06
    depth[out_l] = depth[out_l] + 0.7                                        # far beyond 10 x the median residual
    if R > 100:
        depth[10:60] = gd[10:60] + 0.0078125                                  # ties (exactly representable residuals)
    var = torch.rand(R, generator=gen) * 0.01 + 1e-4
    color, gc = torch.rand(R, 3, generator=gen), torch.rand(R, 3, generator=gen)
    keep = gd > 0
    for with_nan in (False, True):
        dep = depth.clone()
        if with_nan:
            if R < 3:
                continue
            dep[int(torch.nonzero(keep)[1])] = float('nan')
        st = _fake_state(eng, dep, var, color, torch.ones(R, dtype=torch.bool))
        dd, dc, out, scr = eng.empty(R), eng.empty(R, 3), eng.empty(4), eng.empty(R + 8)
        optim.loss_tracker(eng, st, eng.f32(gd), eng.f32(gc), 0.5, True, dd, dc, out, scr, handle_dynamic=False)
        o = out.cpu().numpy()
        if not bool(keep.any()):
            assert o[3] == 0
            continue
        dl, cl = dep[keep].clone().requires_grad_(True), color[keep].clone().requires_grad_(True)
        loss, geo, col, m = H.tracker_loss(dl, var[keep], cl, gd[keep], gc[keep], 0.5, handle_dynamic=False)
        assert o[3] == int(m.sum())
        if with_nan:
            assert o[3] == 0 and o[0] == 0.0 and float(dd.cpu().abs().max()) == 0.0
            continue
        loss.backward()
        assert int(m.sum()) < int(keep.sum()) or R < 20                       # the mask bites
        assert abs(o[0] - loss.item()) <= 2e-5 * abs(loss.item()) and abs(o[1] - geo.item()) <= 2e-5 * abs(geo.item())
        np.testing.assert_allclose(dd.cpu().numpy()[keep.numpy()], dl.grad.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(dc.cpu().numpy()[keep.numpy()], cl.grad.numpy(), atol=1e-7)
        assert float(dd.cpu()[~keep].abs().max() if bool((~keep).any()) else 0.0) == 0.0


@pytest.mark.parametrize('backend', backends())
def test_adam_matches_torch_trajectory(backend):
    """G8: 20 steps, 3 tensors, lr switch geometry->colour, a tensor without gradient in stage 1."""
    eng = make_engine(backend)
    g = load('g8_adam')
    P = [eng.f32(g[k]).clone() for k in ('dec0', 'geo0', 'col0')]
    opt = optim.Adam(eng)
    n_geo = int(g['n_geo_stage'])
    for it in range(20):
        stage_geo = it < n_geo
        lrs = (0.001, 0.03, 0.0) if stage_geo else (0.005, 0.005, 0.005)
        segs = []
        for t, gk in enumerate(('gd', 'gg', 'gc')):
            if t == 2 and stage_geo:
                continue
            segs.append((t, P[t].view(-1), eng.f32(g[gk][it]).view(-1), lrs[t]))
        opt.step(segs)
        for t, k in enumerate(('dec', 'geo', 'col')):
            np.testing.assert_allclose(P[t].cpu().numpy(), g[k][it], rtol=3e-6, atol=1e-7)


@pytest.mark.parametrize('backend', backends())
def test_pose_rays_fwd_bwd(backend):
    eng = make_engine(backend)
    g = load('g6_render_replica_track')
    cam, i, j = tens(g, 'cam', 'i', 'j')
    intr = [float(x) for x in g['intr']]
    R = i.shape[0]
    ro, rd = eng.empty(R, 3), eng.empty(R, 3)
    optim.rays_from_pose(eng, eng.f32(cam), eng.f32(i), eng.f32(j), intr, ro, rd)
    np.testing.assert_allclose(rd.cpu().numpy(), g['rays_d'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ro.cpu().numpy(), g['rays_o'], rtol=0, atol=0)
    gc = eng.empty(7)
    optim.pose_bwd(eng, eng.f32(cam), eng.f32(i), eng.f32(j), intr, eng.f32(g['grad_rays_o']), eng.f32(g['grad_rays_d']), gc)
    assert relerr(gc.cpu(), g['grad_cam']) < 1e-4


@pytest.mark.parametrize('backend', backends())
def test_inside_mask_and_compact(backend):
    eng = make_engine(backend)
    gen = torch.Generator().manual_seed(3)
    for n in (1, 2, 7, 1024, 5000, 12345):
        d = torch.rand(n, generator=gen) * 4
        d[torch.rand(n, generator=gen) < 0.1] = 0.0
        if n > 100:
            d[5] = 1000.0                          # an outlier far beyond 10*median
        pos = d[d > 0]
        mask = eng.empty(n, dtype=torch.uint8)
        thr = eng.empty(1)
        optim.inside_mask(eng, eng.f32(d), mask, thr, eng.empty(n, dtype=torch.int32))
        if pos.numel():
            ref_thr = H.inside_threshold(pos)
            assert float(thr.cpu()) == float(ref_thr)
            ref_mask = (d > 0) & (d <= ref_thr)
        else:
            ref_mask = torch.zeros(n, dtype=torch.bool)
        assert np.array_equal(mask.cpu().numpy().astype(bool), ref_mask.numpy())
        idx = eng.empty(n, dtype=torch.int32)
        cnt = eng.empty(1, dtype=torch.int32)
        optim.compact(eng, mask, idx, cnt)
        k = int(cnt.cpu())
        assert k == int(ref_mask.sum())
        assert np.array_equal(idx.cpu().numpy()[:k], torch.nonzero(ref_mask).reshape(-1).numpy())



@pytest.mark.parametrize('backend', backends())
def test_bucket_copy_roundtrip(backend):
    """lk_bucket_copy (the all-reduce bucket of the data-parallel step): pack = concatenation of weight spans and gathered
    table rows, unpack writes them back - against the torch formulation (index_select / cat / index_copy_)."""
    import ctypes as C
    from loopy_slam_amd import _ffi
    from loopy_slam_amd._ffi import ptr
    eng = make_engine(backend)
    g = torch.Generator().manual_seed(5)
    w = eng.f32(torch.randn(5000, generator=g))
    ta, tb = eng.f32(torch.randn(300, 32, generator=g)), eng.f32(torch.randn(300, 32, generator=g))
    rows = torch.randperm(300, generator=g)[:77].sort().values.to(torch.int32).to(eng.device)
    spans = [(64, 1000), (2048, 1500)]
    segs = (_ffi.CopySeg * 4)()
    for k, (o, n) in enumerate(spans):
        segs[k].data, segs[k].n, segs[k].row_index, segs[k].row_len = ptr(w[o:o + n]), n, None, 1
    for k, t in ((2, ta), (3, tb)):
        segs[k].data, segs[k].n, segs[k].row_index, segs[k].row_len = ptr(t), rows.numel() * 32, ptr(rows), 32
    n = sum(c for _, c in spans) + 2 * rows.numel() * 32
    bucket = eng.zeros(n)
    eng.lib.check(eng.lib.dll.lk_bucket_copy(segs, 4, ptr(bucket), 0, eng.stream), 'lk_bucket_copy')
    ref = torch.cat([w[o:o + c] for o, c in spans] + [t.index_select(0, rows.long()).reshape(-1) for t in (ta, tb)])
    assert torch.equal(bucket.cpu(), ref.cpu())
    # unpack twice the bucket: only the addressed spans / rows change
    w0, ta0 = w.clone(), ta.clone()
    bucket.mul_(2.0)
    eng.lib.check(eng.lib.dll.lk_bucket_copy(segs, 4, ptr(bucket), 1, eng.stream), 'lk_bucket_copy')
    exp_w = w0.clone()
    for o, c in spans:
        exp_w[o:o + c] *= 2.0
    exp_ta = ta0.clone(); exp_ta[rows.long()] *= 2.0
    assert torch.equal(w.cpu(), exp_w.cpu()) and torch.equal(ta.cpu(), exp_ta.cpu())
