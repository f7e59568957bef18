"""Parity of the HIP backward path (lk_render_bwd) against golden gradients captured from the
reference's autograd (tests/golden/g6_*).  Back-ends as in test_forward_parity.py.
The loss gradients d depth / d colour are evaluated on the host exactly as the reference's
Mapper / Tracker losses do (oracle.mapper_loss / tracker_loss on the kernel's outputs);
the fused loss kernels are tested separately in test_optim_parity.py.
Tolerance: max-abs error <= 2e-4 * max|g| per tensor (fp32 reductions in a different order)."""
import numpy as np
import pytest
import torch

from oracle import hotpath as H
from loopy_slam_amd import core
from util import load, tens, weights, CFG, CFG_NAMES, make_engine, backends, relerr
from test_forward_parity import rcfg

torch.set_num_threads(1)
TOL = 2e-4


def setup(eng, name, g, stage, tracker=False, affine=None, color_logits=False, extra_flags=0):
    cfg = rcfg(name)
    W = weights(name)
    dec = core.DecoderBlob(eng).pack(W)
    ro, rd, gd, pos, geo, col = [eng.f32(x) for x in tens(g, 'rays_o', 'rays_d', 'gt_depth', 'pos', 'geo', 'col')]
    knn = core.KnnIndex(eng, capacity=pos.shape[0])
    knn.build(pos)
    st = core.RenderState(eng, ro.shape[0], cfg.S, need_act=True)
    r2 = eng.f32((torch.from_numpy(g['r_query']) ** 2).float()) if CFG[name]['dynamic'] else None
    ng = eng.f32(g['noise_geo'])
    nc = eng.f32(g['noise_col']) if 'noise_col' in g else None
    aff = eng.f32(affine) if affine is not None else None
    core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, dec, stage, tracker=tracker, r2_ray=r2,
                        noise_geo=ng, noise_col=nc, affine=aff, color_logits=color_logits, save_act=True, extra_flags=extra_flags)
    return cfg, dec, st, pos.shape[0], ro.shape[0]


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('unit,geo_dec', ((False, False), (True, False), (True, True)))
@pytest.mark.parametrize('stage', ('geometry', 'color'))
@pytest.mark.parametrize('name', CFG_NAMES)
def test_backward_mapper_golden(backend, name, stage, unit, geo_dec):
    """unit: LK_FLAG_UNIT_LOSS_GRADS as the mapper sets it (its L1 loss gradients are +-1 / +-w): the colour decoder's
    backward then runs on pre-scaled fp16 pieces instead of bf16 pieces - same goldens, same tolerance.
    geo_dec: LK_FLAG_GRAD_GEO_DECODER (mapping.fix_geo_decoder: False) - the geometry decoder's matrices and biases against the
    reference's autograd too."""
    from loopy_slam_amd import _ffi
    eng = make_engine(backend)
    g = load(f'g6_render_{name}_map_{stage}')
    expo = CFG[name]['exposure']
    cfg, dec, st, N, R = setup(eng, name, g, stage, color_logits=expo, extra_flags=_ffi.FLAG_UNIT_LOSS_GRADS if unit else 0)
    # host-side loss gradient on the kernel outputs (Mapper.py:691-720)
    depth = st.depth.cpu().clone().requires_grad_(True)
    color = st.color.cpu().clone().requires_grad_(True)
    gd, gc = tens(g, 'gt_depth', 'gt_color')
    col_for_loss = color
    W = weights(name)
    if expo and stage == 'color':
        aff = H.exposure_affine(W, torch.from_numpy(g['exposure_feat']))
        col_for_loss = torch.sigmoid(color @ aff[:9].reshape(3, 3) + aff[-3:])
    loss, _, _, _ = H.mapper_loss(depth, col_for_loss, st.valid_ray.cpu().bool(), gd, gc, stage, float(g['w_color']))
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    gs = core.GradState(eng, N, R, dec.n, feats=True, weights=True)
    gs.geo_decoder = geo_dec
    core.render_backward(eng, st, gs, eng.f32(depth.grad), eng.f32(color.grad if color.grad is not None else torch.zeros(R, 3)))
    assert relerr(gs.g_geo.cpu(), g['grad_geo']) < TOL
    if 'grad_col' in g:
        assert relerr(gs.g_col.cpu(), g['grad_col']) < TOL
    gW = dec.unpack(gs.g_weights)
    checked = n_geo = 0
    for k, gv in g.items():
        if not k.startswith('gradW.'):
            continue
        nm = k[6:]
        if nm not in gW:
            continue          # exposure MLP lives on the host side
        if nm.startswith('geo_decoder.') and nm != 'geo_decoder.embedder._B':
            if not geo_dec:   # frozen in every reference config (mapping.fix_geo_decoder: True, Mapper.py:537-541): nothing is written
                assert float(gW[nm].abs().max()) == 0.0, nm
                continue
            n_geo += 1
        e = relerr(gW[nm].reshape(gv.shape), gv)
        assert e < TOL, (nm, e)
        checked += 1
    assert checked >= (1 if stage == 'geometry' else 20)
    assert n_geo == (22 if geo_dec else 0)        # 5 x (W, b) + 5 x (U, u) + (w_o, b_o)


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('name', CFG_NAMES)
def test_backward_tracker_golden(backend, name):
    eng = make_engine(backend)
    g = load(f'g6_render_{name}_track')
    W = weights(name)
    aff = None
    if CFG[name]['exposure']:
        ef = torch.from_numpy(g['exposure_feat']).clone().requires_grad_(True)
        aff_t = H.exposure_affine({k: v for k, v in W.items()}, ef)
        aff = aff_t.detach()
    cfg, dec, st, N, R = setup(eng, name, g, 'color', tracker=True, affine=aff)
    depth = st.depth.cpu().clone().requires_grad_(True)
    color = st.color.cpu().clone().requires_grad_(True)
    gd, gc = tens(g, 'gt_depth', 'gt_color')
    loss, _, _, m = H.tracker_loss(depth, st.var.cpu(), color, gd, gc, float(g['w_color']))
    assert np.array_equal(m.numpy(), g['mask'])
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    loss.backward()
    gs = core.GradState(eng, N, R, dec.n, feats=False, weights=False, rays=True, affine=CFG[name]['exposure'])
    core.render_backward(eng, st, gs, eng.f32(depth.grad), eng.f32(color.grad))
    assert relerr(gs.g_rays_d.cpu(), g['grad_rays_d']) < TOL
    assert relerr(gs.g_rays_o.cpu(), g['grad_rays_o']) < TOL
    # chain to the 7-vector pose through the oracle's ray/pose functions
    cam = torch.from_numpy(g['cam']).clone().requires_grad_(True)
    fx, fy, cx, cy = [float(x) for x in g['intr']]
    ro, rd = H.rays_from_uv(torch.from_numpy(g['i']), torch.from_numpy(g['j']), H.quat_to_c2w(cam), fx, fy, cx, cy)
    ((ro * gs.g_rays_o.cpu()).sum() + (rd * gs.g_rays_d.cpu()).sum()).backward()
    assert relerr(cam.grad, g['grad_cam']) < TOL
    if CFG[name]['exposure']:
        aff_t.backward(gs.g_affine.cpu())
        assert relerr(ef.grad, g['grad_exposure_feat']) < TOL


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('N,K,rows,mode', [(128, 128, 333, 0), (128, 168, 100, 1), (32, 128, 640, 2), (3, 128, 77, 0), (128, 40, 64, 1)])
def test_wgrad_single_vs_matmul(backend, N, K, rows, mode):
    """The weight-gradient reduction kernel against a plain matmul (ragged row counts, narrow/wide matrices,
    the three A-operand modes, bias sums)."""
    import ctypes as C
    from loopy_slam_amd._ffi import ptr
    eng = make_engine(backend)
    g = torch.Generator().manual_seed(N * 1000 + K + rows)
    lda = 4 if N == 3 else N + 4
    ldb = K + 8
    B = torch.randn(rows, ldb, generator=g)
    if mode == 2:
        A = torch.randn(rows, lda, generator=g)
        A2 = torch.rand(rows, generator=g)
        Aeff = A2[:, None] * A[:, :N]
        lda2 = 1
    else:
        A = torch.randn(rows, lda, generator=g)
        if N == 3:
            A[:, 3] = 0
        A2 = torch.rand(rows, lda, generator=g) * 0.05
        lda2 = lda
        Aeff = A[:, :N] * (1 - torch.exp(-100 * A2[:, :N])) if mode == 1 else A[:, :N]
    ref = Aeff.double().T @ B[:, :K].double()
    refb = Aeff.double().sum(0)
    ldw = K + 4
    dW, db = eng.zeros(N, ldw), eng.zeros(N)
    Ad, A2d, Bd = eng.f32(A), eng.f32(A2), eng.f32(B)          # keep the device buffers alive across the launch
    rc = eng.lib.dll.lk_wgrad_single(ptr(Ad), lda, mode, ptr(A2d), lda2, ptr(Bd), ldb, N, K, rows,
                                     ptr(dW), ldw, ptr(db), 64, eng.stream)
    eng.lib.check(rc, 'lk_wgrad_single')
    np.testing.assert_allclose(dW.cpu().numpy()[:, :K], ref.numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    assert float(dW.cpu()[:, K:].abs().max()) == 0.0
    np.testing.assert_allclose(db.cpu().numpy(), refb.numpy(), rtol=1e-4, atol=1e-4 * float(refb.abs().max()))


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('name', ('replica', 'tum'))
def test_backward_row_mask(backend, name):
    """grad_row_mask (frustum rows): flagged rows get exactly the unmasked gradient, the others stay untouched;
    decoder gradients are unaffected."""
    eng = make_engine(backend)
    g = load(f'g6_render_{name}_map_color')
    cfg, dec, st, N, R = setup(eng, name, g, 'color')
    gen = torch.Generator().manual_seed(3)
    dd, dc = eng.f32(torch.randn(R, generator=gen)), eng.f32(torch.randn(R, 3, generator=gen))
    full = core.GradState(eng, N, R, dec.n, feats=True, weights=True)
    core.render_backward(eng, st, full, dd, dc)
    mask = (torch.rand(N, generator=gen) < 0.4).to(torch.uint8)
    part = core.GradState(eng, N, R, dec.n, feats=True, weights=True)
    part.row_mask = mask.to(eng.device)
    core.render_backward(eng, st, part, dd, dc)
    m = mask.bool()
    for a, b in ((full.g_geo.cpu(), part.g_geo.cpu()), (full.g_col.cpu(), part.g_col.cpu())):
        scale = float(a.abs().max())
        assert float((a[m] - b[m]).abs().max()) <= 1e-6 * scale            # same terms, atomics may reorder them
        assert float(b[~m].abs().max()) == 0.0 and float(a[~m].abs().max()) > 0.0
    assert relerr(part.g_weights.cpu(), full.g_weights.cpu()) < 1e-6
