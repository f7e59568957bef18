"""Parity of the HIP forward path against the CPU oracle and the golden vectors.

Each test runs on two back-ends:
  emu  (CPU, `-m "not gpu"`): the unmodified .hip sources compiled for the host against
       tests/hipemu — checks the kernels' logic without a GPU;
  hip  (`-m gpu`): the real gfx950 library through the C ABI.
Tolerances: kNN indices / distances / counts bit-exact; rendered depth, colour, variance within
1e-4 relative (fp32), as BASELINE.json's north_star states.
"""
import numpy as np
import pytest
import torch

from oracle import hotpath as H
from loopy_slam_amd import core
from util import load, tens, weights, CFG, CFG_NAMES, make_engine, backends

torch.set_num_threads(1)


def rcfg(name):
    c = CFG[name]
    return core.RenderCfg(S=5, near_surface=c['near_surface'], far_surface=c['far_surface'], near_end=0.3,
                          coef=0.1, min_nn=2, radius_query=0.08, rel_pos=c['rel_pos'], exposure=c['exposure'])


def ocfg(name):
    c = CFG[name]
    return H.RenderCfg(S=5, near_surface=c['near_surface'], far_surface=c['far_surface'], near_end=0.3, coef=0.1,
                       k=8, min_nn=2, radius_query=0.08, rel_pos=c['rel_pos'], exposure=c['exposure'])


# ------------------------------------------------------------------ kNN: bit-exact
@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('name', CFG_NAMES)
def test_knn_matches_oracle_bit_exact(backend, name):
    eng = make_engine(backend)
    g = load(f'g4_interp_{name}')
    p, pos = tens(g, 'p', 'pos')
    if CFG[name]['dynamic']:
        r2 = (torch.from_numpy(g['r_pts']).reshape(-1) ** 2).float()
        r2_o = r2.numpy()
        r2_k = eng.f32(r2)
    else:
        r2_o = np.float32(0.08 ** 2)
        r2_k = float(r2_o)
    od, oi, oc = H.knn_exact(pos.numpy(), p.numpy(), 8, r2_o)
    knn = core.KnnIndex(eng, capacity=pos.shape[0] + 16)
    knn.build(eng.f32(pos))
    d2, idx, cnt = knn.query(eng.f32(p), r2_k)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(d2.cpu().numpy(), od)
    assert np.array_equal(cnt.cpu().numpy(), oc)


@pytest.mark.parametrize('backend', backends())
def test_knn_edge_cases(backend):
    """empty cloud, single point, duplicate points (index tie-break), queries far outside the grid,
    more than 8 points inside the radius, rebuild with more points (append)."""
    eng = make_engine(backend)
    gen = torch.Generator().manual_seed(5)
    knn = core.KnnIndex(eng, capacity=5000, cell_size=0.08)
    q = torch.rand(300, 3, generator=gen) * 2 - 1
    # empty
    knn.build(eng.zeros(0, 3))
    d2, idx, cnt = knn.query(eng.f32(q), 0.01)
    assert (idx.cpu() == -1).all() and (cnt.cpu() == 0).all()
    # duplicates + cluster + outliers
    base = torch.rand(40, 3, generator=gen) * 2 - 1
    pts = torch.cat([base, base[:10], base[:10] + 1e-4, torch.tensor([[50.0, -30.0, 7.0]]),
                     base[:1] + 0.002 * torch.randn(30, 3, generator=gen)])
    qq = torch.cat([q, base, torch.tensor([[1e3, 1e3, 1e3], [-40.0, 2.0, 3.0], [50.0, -30.0, 7.01]])])
    for r2 in (0.0064, 0.0256, 1.0):
        od, oi, oc = H.knn_exact(pts.numpy(), qq.numpy(), 8, np.float32(r2))
        knn.build(eng.f32(pts))
        d2, idx, cnt = knn.query(eng.f32(qq), float(np.float32(r2)))
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(d2.cpu().numpy(), od)
        assert np.array_equal(cnt.cpu().numpy(), oc)
    # append = rebuild with the longer array
    more = torch.cat([pts, torch.rand(700, 3, generator=gen) * 2 - 1])
    knn.build(eng.f32(more))
    assert knn.n == more.shape[0]
    od, oi, oc = H.knn_exact(more.numpy(), qq.numpy(), 8, np.float32(0.0256))
    d2, idx, cnt = knn.query(eng.f32(qq), float(np.float32(0.0256)))
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(cnt.cpu().numpy(), oc)


@pytest.mark.parametrize('backend', backends())
def test_knn_two_phase_search_is_exact(backend):
    """Radii above the cell edge (dynamic query radii up to 0.16 over 0.08-m cells) take the two-phase search of lk_knn_scan_coop:
    the 3 x 3 x 3 cells first, the outer shell only where the list is not yet closed.  Dense sheets (phase 1 closes the list), sparse
    clutter (phase 2 must run), points exactly ON the covered-radius sphere and beyond it, queries on cell boundaries, per-query
    radii on both sides of the switch - indices, distances and counts bit for bit against the brute-force contract."""
    eng = make_engine(backend)
    gen = torch.Generator().manual_seed(11)
    sheet = torch.cat([torch.rand(6000, 2, generator=gen) * 2 - 1, 0.002 * torch.randn(6000, 1, generator=gen)], 1)      # ~1500 points / m^2 at z = 0
    clutter = torch.rand(400, 3, generator=gen) * 2 - 1                                                                  # sparse everywhere else
    q0 = torch.tensor([[0.24, 0.16, 0.4]])                                                                              # a corner of the 0.08 grid lattice
    ring = q0 + torch.nn.functional.normalize(torch.randn(40, 3, generator=gen), dim=1) * torch.linspace(0.075, 0.16, 40)[:, None]
    pts = torch.cat([sheet, clutter, ring])
    q = torch.cat([torch.cat([torch.rand(500, 2, generator=gen) * 1.8 - 0.9, 0.05 * torch.randn(500, 1, generator=gen)], 1),   # near the sheet
                   torch.rand(500, 3, generator=gen) * 2 - 1,                                                               # anywhere
                   q0, q0 + 1e-6, torch.tensor([[0.08, 0.08, 0.08], [-0.16, 0.0, 0.24]])])
    knn = core.KnnIndex(eng, capacity=pts.shape[0], cell_size=0.08)
    knn.build(eng.f32(pts))
    # the grid's origin is the cloud's min corner: shift the lattice queries onto its cell boundaries
    r_per = 0.04 + 0.26 * torch.rand(q.shape[0], generator=gen)
    for r2 in ((r_per ** 2).float(), np.float32(0.16 ** 2), np.float32(0.0841 ** 2), np.float32(0.084 ** 2), np.float32(0.3 ** 2)):
        per = torch.is_tensor(r2)
        od, oi, oc = H.knn_exact(pts.numpy(), q.numpy(), 8, r2.numpy() if per else r2)
        d2, idx, cnt = knn.query(eng.f32(q), eng.f32(r2) if per else float(r2))
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(d2.cpu().numpy(), od)
        assert np.array_equal(cnt.cpu().numpy(), oc)
    # both outcomes occur: lists closed inside one cell edge, and lists that needed the shell
    full_near = (od[:, 7] <= np.float32(0.0799 ** 2)).sum()
    assert 100 < full_near < q.shape[0] - 100


@pytest.mark.parametrize('backend', backends())
def test_knn_append_equals_build(backend):
    """lk_knn_append (index.add of add_neural_points): appending in three pieces answers every query exactly as one build over
    the whole array - indices refer to the concatenated order."""
    eng = make_engine(backend)
    gen = torch.Generator().manual_seed(9)
    pts = torch.rand(3000, 3, generator=gen) * 2 - 1
    q = torch.rand(500, 3, generator=gen) * 2 - 1
    a = core.KnnIndex(eng, capacity=4000, cell_size=0.08)
    a.build(eng.f32(pts))
    b = core.KnnIndex(eng, capacity=4000, cell_size=0.08)
    b.build(eng.f32(pts[:1200]))
    b.append(eng.f32(pts[1200:1201]))
    b.append(eng.f32(pts[1201:2500]))
    b.append(eng.f32(pts[2500:]))
    assert b.n == 3000
    for r2 in (0.0064, 0.04):
        da, ia, ca = a.query(eng.f32(q), r2)
        db, ib, cb = b.query(eng.f32(q), r2)
        assert torch.equal(ia, ib) and torch.equal(da, db) and torch.equal(ca, cb)
    od, oi, oc = H.knn_exact(pts.numpy(), q.numpy(), 8, np.float32(0.04))
    assert np.array_equal(ib.cpu().numpy(), oi)


# ------------------------------------------------------------------ weights blob round trip
@pytest.mark.parametrize('backend', backends())
def test_weight_blob_roundtrip(backend):
    eng = make_engine(backend)
    W = weights('replica')
    blob = core.DecoderBlob(eng).pack(W)
    back = blob.unpack()
    n = 0
    for k, v in back.items():
        assert torch.equal(v.reshape(W[k].shape), W[k]), k
        n += 1
    assert n >= 50
    # padding stays zero
    assert float(blob.blob.abs().sum().cpu()) == pytest.approx(sum(float(W[k].abs().sum()) for k in back), rel=1e-5)


# ------------------------------------------------------------------ end-to-end forward vs golden + oracle
def run_forward(eng, name, g, stage, tracker=False, affine=None, color_logits=False, stats_chunk=None, W=None):
    cfg = rcfg(name)
    W = weights(name) if W is None else W
    blob = core.DecoderBlob(eng).pack(W)
    ro, rd, gd, pos, geo, col = [eng.f32(x) for x in tens(g, 'rays_o', 'rays_d', 'gt_depth', 'pos', 'geo', 'col')]
    knn = core.KnnIndex(eng, capacity=pos.shape[0])
    knn.build(pos)
    st = core.RenderState(eng, ro.shape[0], cfg.S)
    r2 = eng.f32((torch.from_numpy(g['r_query']) ** 2).float()) if CFG[name]['dynamic'] else None
    ng = eng.f32(g['noise_geo'])
    nc = eng.f32(g['noise_col']) if 'noise_col' in g else None
    aff = eng.f32(affine) if affine is not None else None
    core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, stage, tracker=tracker, r2_ray=r2,
                        noise_geo=ng, noise_col=nc, affine=aff, color_logits=color_logits, stats_chunk=stats_chunk)
    return st


def check_outputs(st, g, has_color=True):
    assert np.array_equal(st.valid_ray.cpu().numpy().astype(bool), g['valid_ray'])
    np.testing.assert_allclose(st.depth.cpu().numpy(), g['depth'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st.var.cpu().numpy(), g["var"], rtol=2e-4, atol=1e-9)       # the at-size bound (tests/test_parity_at_size.py: TOL_VAR)
    if has_color:
        np.testing.assert_allclose(st.color.cpu().numpy(), g['color'], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('stage', ('geometry', 'color'))
@pytest.mark.parametrize('name', CFG_NAMES)
def test_forward_mapper_golden(backend, name, stage):
    eng = make_engine(backend)
    g = load(f'g6_render_{name}_map_{stage}')
    st = run_forward(eng, name, g, stage, color_logits=CFG[name]['exposure'])
    check_outputs(st, g)


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('name', CFG_NAMES)
def test_forward_tracker_golden(backend, name):
    eng = make_engine(backend)
    g = load(f'g6_render_{name}_track')
    aff = None
    if CFG[name]['exposure']:
        aff = H.exposure_affine(weights(name), torch.from_numpy(g['exposure_feat']))
    st = run_forward(eng, name, g, 'color', tracker=True, affine=aff)
    check_outputs(st, g)


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('name', CFG_NAMES)
def test_forward_img_zero_depth_golden(backend, name):
    eng = make_engine(backend)
    g = load(f'g6_render_{name}_img')
    aff = None
    if CFG[name]['exposure']:
        aff = H.exposure_affine(weights(name), torch.from_numpy(g['exposure_feat']))
    st = run_forward(eng, name, g, 'color', affine=aff)
    check_outputs(st, g)


@pytest.mark.parametrize('backend', backends())
def test_forward_intermediates_vs_oracle(backend):
    """z, neighbour lists, weights, interpolated features and raw decoder outputs against the oracle."""
    eng = make_engine(backend)
    name = 'replica'
    g = load(f'g6_render_{name}_map_color')
    st = run_forward(eng, name, g, 'color')
    W = weights(name)
    ro, rd, gd, pos, geo, col = tens(g, 'rays_o', 'rays_d', 'gt_depth', 'pos', 'geo', 'col')
    with torch.no_grad():
        o = H.render_batch(ocfg(name), ro, rd, gd, pos, geo, col, W, 'color',
                           noise_geo=torch.from_numpy(g['noise_geo']), noise_col=torch.from_numpy(g['noise_col']))
    assert np.array_equal(st.z.cpu().numpy(), o['z'].numpy())                       # bit-exact sample depths
    assert np.array_equal(st.nbr_idx.cpu().numpy(), o['idx'].numpy())               # bit-exact neighbours
    assert np.array_equal(st.nbr_count.cpu().numpy(), o['count'].numpy())
    raw = st.raw.cpu().numpy()
    np.testing.assert_allclose(raw[:, 3], o['occ'].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(raw[:, :3], o['rgb'].numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('scale,wscale', ((1e-3, 1.0), (30.0, 1.0), (10.0, 3.0), (1.0, 0.1)))
def test_forward_operand_range(backend, scale, wscale):
    """The forward products run on fp16 pieces (lk_common.h: fp16x3): feature tables 30x larger (activations of a few
    hundred), 1000x smaller (low pieces deep in the fp16 subnormals), larger / smaller decoder matrices (every weight matrix of both
    decoders and the rel-pos MLP times 3 with 10x features: activations of ~1e4, a sixth of fp16's range; times 0.1) still match the
    oracle.  The hard limit is fp16's 65 504 for any single operand (feature, activation or weight)."""
    eng = make_engine(backend)
    name = 'replica'
    g = dict(load(f'g6_render_{name}_map_color'))
    g['geo'], g['col'] = g['geo'] * np.float32(scale), g['col'] * np.float32(scale)
    W = {k: (v * wscale if (v.dim() == 2 and not k.endswith('_B')) else v.clone()) for k, v in weights(name).items()}
    st = run_forward(eng, name, g, 'color', W=W)
    ro, rd, gd, pos, geo, col = tens(g, 'rays_o', 'rays_d', 'gt_depth', 'pos', 'geo', 'col')
    with torch.no_grad():
        o = H.render_batch(ocfg(name), ro, rd, gd, pos, geo, col, W, 'color',
                           noise_geo=torch.from_numpy(g['noise_geo']), noise_col=torch.from_numpy(g['noise_col']))
    np.testing.assert_allclose(st.depth.cpu().numpy(), o['depth'].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st.color.cpu().numpy(), o['color'].numpy(), rtol=1e-4, atol=2e-5)
    raw = st.raw.cpu().numpy()
    amax = float(np.abs(o['occ'].numpy()).max())
    np.testing.assert_allclose(raw[:, 3], o['occ'].numpy(), rtol=1e-4, atol=1e-4 * max(1.0, scale, 1e-2 * amax))
