"""Drop-in API (loopy_slam_amd.slam): reference names / signatures over the kernels.
Miniature synthetic room so the emulator back-end finishes in seconds; the gpu back-end runs the same."""
import copy

import os

import numpy as np
import pytest
import torch

from oracle import hotpath as H
from loopy_slam_amd import slam, config, core
from util import make_engine, backends

torch.set_num_threads(1)


def mini_cfg():
    cfg = config.load_config('configs/Synthetic/room.yaml', 'configs/point_slam.yaml')
    cfg = copy.deepcopy(cfg)
    cfg['cam'].update(H=24, W=32, fx=26.0, fy=26.0, cx=15.5, cy=11.5)
    cfg['tracking'].update(ignore_edge_W=2, ignore_edge_H=2, pixels=48, iters=3)
    cfg['mapping'].update(pixels=64, pixels_adding=400, iters=3, iters_first=6, geo_iter_first=2, every_frame=2, keyframe_every=2,
                          mapping_window_size=4)
    cfg['pointcloud'].update(radius_add=0.12, radius_query=0.24, radius_min=0.06)     # coarse image -> coarser cloud
    cfg['data']['n_frames'] = 4
    return cfg


@pytest.mark.parametrize('backend', backends())
def test_point_slam_runs_and_tracks(backend):
    eng = make_engine(backend)
    cfg = mini_cfg()
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    est, gt = ps.run()
    assert ps.npc.pts_num() > 300 and ps.npc.pts_num() % 3 == 0
    assert torch.isfinite(est).all()
    # poses stay near the ground truth (tiny motion between synthetic frames, few iterations)
    assert float((est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.1
    log = ps.mapper.last_log.cpu()
    assert torch.isfinite(log).all() and log[:, 3].min() > 0
    assert float(log[-1, 1]) < float(log[0, 1]) * 1.5          # geometry loss does not blow up


@pytest.mark.parametrize('backend', backends())
def test_renderer_autograd_bridge_matches_oracle(backend):
    """Renderer.render_batch_ray with the reference signature, gradients through torch autograd."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    idx, color, depth, c2w = ps.frame_reader[0]
    ps.tracker.track_frame(0, color, depth, c2w)
    ps.mapper.map_frame(0, color, depth, c2w, cur_c2w=c2w)
    npc, dec = ps.npc, ps.shared_decoders
    g = torch.Generator().manual_seed(0)
    ii = torch.randint(2, 30, (40,), generator=g).float()
    jj = torch.randint(2, 22, (40,), generator=g).float()
    ro, rd = H.rays_from_uv(ii, jj, c2w.cpu(), 26.0, 26.0, 15.5, 11.5)
    gd = depth.cpu()[jj.long(), ii.long()]
    geo = npc.get_geo_feats().clone().requires_grad_(True)
    col = npc.get_col_feats().clone().requires_grad_(True)
    ps.renderer.sigmoid_coefficient = 0.1
    d, u, c, valid = ps.renderer.render_batch_ray(npc, dec, eng.f32(rd), eng.f32(ro), eng.device, 'color', gt_depth=eng.f32(gd),
                                                   npc_geo_feats=geo, npc_col_feats=col)
    loss = (d * 1.3).sum() + (c * c).sum() + 0.5 * u.sum()
    loss.backward()
    # oracle
    W = {k: v for k, v in dec.dec.unpack().items()}
    go = npc.get_geo_feats().cpu().clone().requires_grad_(True)
    co = npc.get_col_feats().cpu().clone().requires_grad_(True)
    ocfg = H.RenderCfg(radius_query=cfg['pointcloud']['radius_query'], rel_pos=cfg['model']['encode_rel_pos_in_col'])
    out = H.render_batch(ocfg, ro, rd, gd, npc.cloud_pos().cpu(), go, co, W, 'color')
    lo = (out['depth'] * 1.3).sum() + (out['color'] * out['color']).sum() + 0.5 * out['var'].sum()
    lo.backward()
    np.testing.assert_allclose(d.detach().cpu().numpy(), out['depth'].detach().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(c.detach().cpu().numpy(), out['color'].detach().numpy(), rtol=1e-4, atol=2e-5)
    assert np.array_equal(valid.cpu().numpy(), out['valid_ray'].numpy())
    sc = float(go.grad.abs().max())
    assert float((geo.grad.cpu() - go.grad).abs().max()) < 2e-4 * sc
    assert float((col.grad.cpu() - co.grad).abs().max()) < 2e-4 * float(co.grad.abs().max())


@pytest.mark.parametrize('backend', backends())
def test_npc_add_dedup_and_nicer_state_dict(backend):
    eng = make_engine(backend)
    cfg = mini_cfg()
    npc = slam.NeuralPointCloud(cfg, eng=eng, capacity=64)            # forces a grow
    g = torch.Generator().manual_seed(1)
    ro = torch.zeros(200, 3)
    rd = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=1)
    gd = 1 + torch.rand(200, generator=g)
    gd[:10] = 0
    n1 = npc.add_neural_points(eng.f32(ro), eng.f32(rd), eng.f32(gd), eng.f32(torch.rand(200, 3, generator=g)))
    assert n1 == 190 and npc.pts_num() == 570
    n2 = npc.add_neural_points(eng.f32(ro), eng.f32(rd), eng.f32(gd), eng.f32(torch.rand(200, 3, generator=g)))
    assert n2 == 0 and npc.pts_num() == 570                            # every location already has a neighbour within radius_add
    D, I, cnt = npc.find_neighbors_faiss(npc.cloud_pos()[:50], step='query')
    assert I.dtype == torch.int64 and (I[:, 0].cpu() == torch.arange(50)).all() and (D[:, 0].cpu() == 0).all()
    dec = slam.NICER(cfg, eng=eng)
    sd = dec.state_dict()
    assert 'geo_decoder.pts_linears.3.weight' in sd and tuple(sd['geo_decoder.pts_linears.3.weight'].shape) == (32, 125)
    assert 'color_decoder.embedder._B' not in sd and len(sd) == 55
    sd2 = {k: v + 0.25 for k, v in sd.items()}
    dec.load_state_dict(sd2)
    for k, v in dec.state_dict().items():
        assert torch.allclose(v, sd2[k]), k
    dec.geo_decoder.load_state_dict({'output_linear.bias': torch.tensor([3.0])}, strict=False)
    assert float(dec.state_dict()['geo_decoder.output_linear.bias']) == 3.0
    raw, ray_mask, point_mask = dec.forward(npc.cloud_pos()[:20], npc, 'color', npc.get_geo_feats(), npc.get_col_feats(), pts_num=5)
    assert raw.shape == (20, 4) and point_mask.shape == (20,) and ray_mask.shape == (4,)


@pytest.mark.parametrize('backend', backends())
def test_point_slam_dynamic_radius_and_gradient_sampling(backend):
    """TUM / ScanNet style configuration: per-pixel radii from the colour gradient (insertion + every render of the
    mapping window + tracking) and tracking rays drawn from the high-gradient pixel pool."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['use_dynamic_radius'] = True
    cfg['tracking']['sample_with_color_grad'] = True
    cfg['pointcloud'].update(radius_add_max=0.08, radius_add_min=0.02, radius_query_ratio=2, color_grad_threshold=0.15)
    cfg['cam']['crop_edge'] = 1            # both configs crop the frames (tum.yaml / scannet.yaml): 22 x 30 images, cx, cy shifted
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    assert (ps.H, ps.W) == (22, 30)
    est, gt = ps.run()
    assert ps.npc.pts_num() > 300 and torch.isfinite(est).all()
    assert float((est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.1
    kf = ps.mapper.keyframe_dict[0]
    r2 = kf['r2_query']
    assert r2 is not None and tuple(r2.shape) == tuple(kf['depth'].shape)
    assert float(r2.max()) <= np.float32(0.16 ** 2) and float(r2.min()) >= np.float32(0.04 ** 2)
    assert torch.isfinite(ps.mapper.last_log.cpu()).all() and torch.isfinite(ps.tracker.last_log.cpu()).all()


@pytest.mark.parametrize('backend', backends())
def test_checkpoint_roundtrip_reference_format(backend, tmp_path):
    """Logger writes the reference's `.tar` layout (src/utils/Logger.py:20-65: same keys, positions as lists, decoder
    state_dict with the reference's names) and a second system resumes from it: same map, same renders."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    ps.run(n_frames=3)
    lg = slam.Logger(cfg, None, ps.mapper, ckptsdir=str(tmp_path))
    path = lg.log(2, ps.mapper.keyframe_dict, ps.mapper.keyframe_list, npc=ps.npc, last_log=True)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    for k in ('geo_feats', 'col_feats', 'cloud_pos', 'pts_num', 'input_pos', 'input_rgb', 'input_normal', 'input_normal_cartesian',
              'decoder_state_dict', 'gt_c2w_list', 'estimate_c2w_list', 'keyframe_list', 'keyframe_dict', 'selected_keyframes', 'idx',
              'fragments', 'exposure_feat_all'):
        assert k in ck, k
    assert isinstance(ck['cloud_pos'], list) and len(ck['cloud_pos']) == ck['pts_num'] == ps.npc.pts_num()
    assert 'geo_decoder.pts_linears.0.weight' in ck['decoder_state_dict'] and ck['idx'] == 2
    ps2 = slam.Point_SLAM(cfg, None, eng=eng)
    assert slam.Logger.load(path, ps2) == 2
    assert torch.equal(ps2.npc.cloud_pos().cpu(), ps.npc.cloud_pos().cpu())
    assert torch.equal(ps2.npc.get_geo_feats().cpu(), ps.npc.get_geo_feats().cpu())
    assert len(ps2.mapper.keyframe_dict) == len(ps.mapper.keyframe_dict)
    idx, color, depth, c2w = ps.frame_reader[1]
    d1, u1, c1 = ps.renderer.render_img(ps.npc, ps.shared_decoders, c2w, eng.device, 'color', gt_depth=depth)
    d2, u2, c2 = ps2.renderer.render_img(ps2.npc, ps2.shared_decoders, c2w, eng.device, 'color', gt_depth=depth)
    assert torch.equal(d1.cpu(), d2.cpu()) and torch.equal(c1.cpu(), c2.cpu())


@pytest.mark.parametrize('backend', backends())
def test_exposure_features_are_checkpointed(backend, tmp_path):
    """model.encode_exposure: the mapper keeps the optimised exposure feature of every optimize_map call (Mapper.py:800, 827),
    Mapper.run hands the list to the Logger (Mapper.py:1028-1031), and Logger.load restores it (Mapper.py:394, 1111;
    get_mesh_tsdf_fusion.py:44-46).  A keyframe is not inserted for a frame whose ground-truth pose is not finite (Mapper.py:982)."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['model']['encode_exposure'] = True
    cfg['mapping']['ckpt_freq'] = 2
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    ps.mapper.logger = slam.Logger(cfg, None, ps.mapper, ckptsdir=str(tmp_path))
    ps.run()
    n_mapped = sum(1 for i in range(ps.n_img) if i == 0 or i % ps.mapper.every_frame == 0 or i == ps.n_img - 1)
    xa = ps.mapper.exposure_feat_all
    assert len(xa) >= n_mapped and all(tuple(e.shape) == (cfg['model']['exposure_dim'],) for e in xa)
    assert float(torch.stack(xa).abs().max()) > 0                          # the feature moved off its zero start
    path = os.path.join(str(tmp_path), '{:05d}.tar'.format(ps.n_img - 1))
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert ck['exposure_feat_all'] is not None and torch.equal(ck['exposure_feat_all'], torch.stack(xa))
    ps2 = slam.Point_SLAM(cfg, None, eng=eng)
    slam.Logger.load(path, ps2)
    assert len(ps2.mapper.exposure_feat_all) == len(xa) and torch.equal(ps2.mapper.exposure_feat_all[-1], xa[-1])
    assert torch.equal(ps2.exposure_feat.cpu(), xa[-1])
    # non-finite ground-truth pose: the frame is mapped, but never becomes a keyframe
    ps3 = slam.Point_SLAM(mini_cfg(), None, eng=eng)
    idx, color, depth, c2w = ps3.frame_reader[0]
    ps3.tracker.track_frame(0, color, depth, c2w)
    bad = c2w.clone(); bad[0, 3] = float('-inf')
    ps3.mapper.map_frame(0, color, depth, bad, cur_c2w=c2w)
    assert ps3.mapper.keyframe_list == []


@pytest.mark.parametrize('backend', backends())
def test_sample_near_pcl(backend):
    """NeuralPointCloud.sample_near_pcl against the oracle restatement (probes through the device kNN)."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    npc = slam.NeuralPointCloud(cfg, eng=eng, capacity=4096)
    g = torch.Generator().manual_seed(3)
    wall = torch.cat([torch.rand(1500, 2, generator=g) * 4 - 2, torch.full((1500, 1), 2.0)], 1)       # a wall at z = 2
    wall2 = torch.cat([torch.rand(600, 2, generator=g) * 4 - 2, torch.full((600, 1), 3.1)], 1)       # a second one behind it
    npc._grow(2100); npc._pos[:2100] = eng.f32(torch.cat([wall, wall2])); npc.n = 2100
    npc.knn.build(npc._pos[:2100])
    ro = torch.zeros(40, 3)
    rd = torch.cat([torch.rand(40, 2, generator=g) * 0.8 - 0.4, torch.ones(40, 1)], 1)
    rd[:5, 2] = -1.0                                                                                   # looking away: invalid
    z, inv = npc.sample_near_pcl(eng.f32(ro), eng.f32(rd), 0.3, 4.0, 5)
    zr, invr = H.sample_near_pcl(ro, rd, 0.3, 4.0, 5, npc.cloud_pos().cpu().numpy(), npc.radius_query)
    assert np.array_equal(inv.cpu().numpy(), invr) and invr[:5].all() and not invr.all()
    assert np.array_equal(z.cpu().numpy(), zr)


@pytest.mark.parametrize('backend', backends())
def test_point_slam_exposure_config(backend):
    """ScanNet style model (encode_exposure, no rel-pos): the tracker optimises the frame's exposure feature, keyframes keep
    theirs and the mapper optimises them together with mlp_exposure."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['model']['encode_exposure'] = True
    cfg['model']['encode_rel_pos_in_col'] = False
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    w0 = ps.shared_decoders.mlp_exposure[2].bias.detach().clone()
    est, gt = ps.run()
    assert torch.isfinite(est).all() and float((est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.1
    kfs = ps.mapper.keyframe_dict
    assert len(kfs) >= 2 and all(kf['exposure_feat'] is not None and kf['exposure_feat'].shape == (8,) for kf in kfs)
    assert float(ps.exposure_feat.abs().max()) > 0 and float(kfs[0]['exposure_feat'].detach().abs().max()) > 0       # both moved off zero
    assert float((ps.shared_decoders.mlp_exposure[2].bias.detach() - w0).abs().max()) > 0
    assert torch.isfinite(ps.mapper.last_log.cpu()).all() and torch.isfinite(ps.tracker.last_log.cpu()).all()


@pytest.mark.parametrize('backend', backends())
def test_add_points_schedule_matches_oracle(backend):
    """Mapper.optimize_map's insertion passes (src/Mapper.py:421-482): first-frame count scaling, non-overlapping area,
    1000 overlap samples, colour-gradient pixels with radius_min - accepted counts per pass, frame_pts_add, the grown cloud
    (bit for bit) and the iteration count derived from it (Mapper.py:572-574) against the oracle restatement on the SAME
    pixel draws."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['mapping'].update(pixels_adding=300, pixels_based_on_color_grad=40, iters=6)
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    mp, pc = ps.mapper, cfg['pointcloud']
    ocfg = dict(pixels_adding=300, pixels_grad=40, radius_add=pc['radius_add'], radius_min=pc['radius_min'],
                near=pc['near_end_surface'], far=pc['far_end_surface'], filter_before=True)
    intr = (ps.fx, ps.fy, ps.cx, ps.cy)
    prev = None
    for idx in (0, 2):
        _, color, depth, c2w = ps.frame_reader[idx]
        draws = mp.draw_add_pixels(idx, depth, color)
        if idx == 0:
            assert draws['main'].numel() == H.first_frame_add_count(depth.cpu(), 300) and 300 <= draws['main'].numel() <= 900
        # the gradient pool draw: n distinct pixels among the 5n largest gradient magnitudes, sorted (common.py:175-196)
        g = H.color_grad_mag(color.cpu().numpy())
        pool = set(H.top_grad_pixels(g, 5 * 40, (0, ps.H, 0, ps.W)).tolist())
        gd_ = draws['grad'].cpu().tolist()
        assert len(gd_) == 40 == len(set(gd_)) and gd_ == sorted(gd_) and set(gd_) <= pool
        before = ps.npc.cloud_pos().cpu().clone()
        total, counts = mp.add_points_for_frame(idx, color, depth, c2w, draws=draws)
        o_total, o_counts, o_cloud = H.add_points_schedule(idx, depth.cpu(), color.cpu(), c2w.cpu(), prev, before, intr, ocfg,
                                                           {k: v.cpu() for k, v in draws.items()})
        assert counts == o_counts and total == o_total, (idx, counts, o_counts)
        assert len(counts) == (2 if idx == 0 else 3) and total > 0
        # same points; the rays of the draws are torch expressions on the engine's device (bit-identical to the oracle's on the CPU
        # back-end, a rounding apart on the GPU where torch sums the three products of a direction in another order)
        if backend == 'emu':
            assert torch.equal(ps.npc.cloud_pos().cpu(), o_cloud)
        np.testing.assert_allclose(ps.npc.cloud_pos().cpu().numpy(), o_cloud.numpy(), rtol=0, atol=2e-6)
        mp.prev_c2w = c2w.clone()
        prev = c2w.cpu()
    # the iteration count of a mapped frame follows from what was added (and is clipped on both sides)
    _, color, depth, c2w = ps.frame_reader[3]
    n0 = ps.npc.pts_num()
    mp.optimize_map(6, 3, color, depth, c2w, mp.keyframe_dict, mp.keyframe_list, c2w)
    assert mp.last_frame_pts_add == (ps.npc.pts_num() - n0) // 3 == sum(mp.last_add_counts)
    assert mp.last_num_joint_iters == H.mapping_iterations(6, mp.last_frame_pts_add, cfg['mapping']['min_iter_ratio'])
    assert tuple(mp.last_log.shape) == (mp.last_num_joint_iters, 4)


@pytest.mark.parametrize('backend', backends())
def test_tracking_depth_limit_without_gradient_sampling(backend):
    """tracking.depth_limit with uniform pixel draws (Tracker.py:142-146, common.py:249-252): readings of 5 m and beyond are dropped
    like missing ones - the loop's masked-ray count equals the count over the limited image."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['tracking'].update(depth_limit=True, sample_with_color_grad=False, iters=2)
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    _, color, depth, c2w = ps.frame_reader[0]
    ps.tracker.track_frame(0, color, depth, c2w)
    ps.mapper.map_frame(0, color, depth, c2w, cur_c2w=c2w)
    _, color1, depth1, c2w1 = ps.frame_reader[1]
    ps.tracker.track_frame(1, color1, depth1, c2w1)  # (the first two frames keep the given pose, Tracker.py:297: frame 2 is the first tracked one)
    _, color1, depth1, c2w1 = ps.frame_reader[2]
    far = depth1.clone()
    far[::2, ::3] = 6.5                              # a third of a half of the pixels beyond the limit
    counts = {}
    for name, img in (('limited', far), ('plain', depth1)):
        ps.tracker.gen.manual_seed(77)
        ps.tracker.track_frame(2, color1, img, c2w1)
        counts[name] = ps.tracker.last_log[:, 3].cpu()
    n_px = cfg['tracking']['pixels']
    assert float(counts['limited'].max()) < 0.95 * float(counts['plain'].min()) and float(counts['limited'].min()) > 0.5 * n_px
    # without the option the far readings take part (and the inside mask / outlier mask may or may not drop them)
    cfg2 = mini_cfg()
    cfg2['tracking'].update(depth_limit=False, iters=2)
    assert slam.Point_SLAM(cfg2, None, eng=eng).tracker.depth_limit is False


@pytest.mark.parametrize('backend', backends())
def test_fix_geo_decoder_false_in_the_mapper(backend):
    """mapping.fix_geo_decoder: False (Mapper.py:524-526): the geometry decoder's own matrices are parameters of the mapper's decoder group -
    they move, the run stays on track (with the default they receive no gradient at all: test_backward_mapper_golden)."""
    eng = make_engine(backend)
    moved = {}
    for fix in (False,):
        cfg = mini_cfg()
        cfg['mapping'].update(fix_geo_decoder=fix, iters_first=4, iters=2)
        cfg['data']['n_frames'] = 3
        ps = slam.Point_SLAM(cfg, None, eng=eng)
        w0 = {k: v.clone() for k, v in ps.shared_decoders.dec.unpack().items() if k.startswith('geo_decoder.')}
        est, gt = ps.run()
        w1 = ps.shared_decoders.dec.unpack()
        moved[fix] = {k: float((w1[k] - w0[k]).abs().max()) for k in w0}
        assert torch.isfinite(est).all() and float((est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.1
    for k in ('geo_decoder.pts_linears.0.weight', 'geo_decoder.pts_linears.3.weight', 'geo_decoder.fc_c.2.weight', 'geo_decoder.output_linear.weight',
              'geo_decoder.pts_linears.4.bias'):
        assert moved[False][k] > 1e-5, (k, moved[False][k])


@pytest.mark.parametrize('backend', backends())
def test_bundle_adjustment_in_the_mapper(backend):
    """mapping.BA: True (Mapper.py:541-566, 782-797, 957-964): off until the run holds more than four keyframes; then the window's
    poses (but the oldest keyframe's) are optimised with the map, written back into the keyframes, and the mapped frame's estimate is
    replaced by the optimised pose.  handle_dynamic: False rides along (the tracker's median mask)."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['mapping'].update(BA=True, BA_cam_lr=0.002, every_frame=1, keyframe_every=1, iters=10, color_refine=False, mapping_window_size=4,
                          keyframe_selection_method='global')
    cfg['tracking'].update(handle_dynamic=False)
    cfg['data']['n_frames'] = 6                      # five keyframes after frame 4: the sixth frame is mapped with BA
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    seen = []
    orig = ps.mapper.optimize_map

    def spy(num_joint_iters, idx, *a, **k):
        kd = a[3]                                   # keyframe_dict
        before = [d['est_c2w'].clone() for d in kd]
        # (the frames before BA starts only have to leave keyframes behind: three iterations each keep the emulator run short)
        r = orig(num_joint_iters if ps.mapper.BA or idx == 0 else 3, idx, *a, **k)
        moved = [bool((d['est_c2w'] != b).any()) for d, b in zip(kd, before)]
        seen.append(dict(idx=idx, ba=ps.mapper.BA, n_kf=len(before), moved=moved, ret=r, cur=a[5].clone()))
        return r
    ps.mapper.optimize_map = spy
    est, gt = ps.run()
    assert [c['ba'] for c in seen] == [c['n_kf'] > 4 for c in seen] and any(c['ba'] for c in seen) and not seen[0]['ba']
    for c in seen:
        if not c['ba']:
            assert c['ret'] is None and not any(c['moved'])
            continue
        assert sum(c['moved']) == 2                                          # window of 3 keyframes: the oldest of them stays fixed
        assert c['moved'].index(True) == c['n_kf'] - 2
        assert c['ret'] is not None and tuple(c['ret'].shape) == (4, 4)
        d = float((c['ret'][:3, 3] - c['cur'][:3, 3]).abs().max())
        assert 0.0 < d < 0.05                                                 # the frame's pose moved, by Adam-sized steps
        assert torch.allclose(est[c['idx']], c['ret'].cpu(), atol=0)
        R3 = c['ret'][:3, :3]
        assert float((R3 @ R3.T - torch.eye(3, device=R3.device)).abs().max()) < 1e-5
    assert torch.isfinite(est).all() and float((est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.1


@pytest.mark.parametrize('backend', backends())
def test_final_refinement_optimises_the_whole_map(backend):
    """mapping.color_refine (Mapper.py:884-897): on the last frame every row of the map is trainable (no frustum selection),
    the colour decoder is frozen, ten times the iterations in five optimize_map calls, no points are added."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['mapping'].update(iters=2, color_refine=True, pixels=96, iters_first=8, geo_iter_first=3)
    # segments (Mapper.py:338-345, neural_point.py:1317-1326): the synthetic camera moves 2-7 mm per frame - with a 5 mm threshold the mapped
    # frame 2 opens a segment behind frame 0's, and the refinement's window is the segments that exist when the last frame is mapped + that frame
    cfg['mapping'].update(segment_strategy='rot_trans', segment_rel_trans=0.005, segment_rot_cos=0.0)
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    calls = []
    orig = ps.mapper.optimize_map

    def spy(num_joint_iters, idx, *a, **k):
        n0 = ps.npc.pts_num()
        geo0 = ps.npc.get_geo_feats().clone()
        w0 = ps.shared_decoders.dec.blob.clone()
        r = orig(num_joint_iters, idx, *a, **k)
        moved = (ps.npc.get_geo_feats()[:n0] != geo0).any(1)
        calls.append(dict(idx=idx, iters=num_joint_iters, refine=k.get('color_refine', False), added=ps.npc.pts_num() - n0, window=list(ps.mapper.last_window_idx),
                          frac_moved=float(moved.float().mean()), w_moved=int((ps.shared_decoders.dec.blob != w0).sum())))
        return r
    ps.mapper.optimize_map = spy
    ps.run()
    last = [c for c in calls if c['idx'] == 3]
    assert len(last) == 5 and all(c['refine'] and c['iters'] == 2 * 10 // 5 and c['added'] == 0 for c in last)
    step = [float((ps.estimate_c2w_list[k + 1][:3, 3] - ps.estimate_c2w_list[k][:3, 3]).norm()) for k in range(3)]
    seg = [s['idx'] for s in ps.mapper.segments]
    assert seg[0] == 0 and 2 <= len(seg) <= 3 and all(0.001 < x < 0.02 for x in step), (seg, step)
    # the refinement's window: one keyframe per segment that existed when the last frame was mapped, plus that frame
    assert all(c['window'] == [i for i in seg if i < 3] + [3] for c in last), (last[0]['window'], seg)
    ps.mapper.segments = ps.mapper.segments[:1]            # a caller's own segmentation: one segment -> {0} + the current frame
    ps.mapper.keyframe_selection_method = 'segments'
    ps.mapper.optimize_map(2, 3, *ps.frame_reader[3][1:3], ps.frame_reader[3][3], ps.mapper.keyframe_dict, ps.mapper.keyframe_list,
                           ps.estimate_c2w_list[3].to(eng.device), color_refine=True)
    assert ps.mapper.last_window_idx == [0, 3]
    # rows far outside the last frustum moved too; only the two embedding matrices of the decoders may change (fix_color_decoder)
    assert min(c['frac_moved'] for c in last) > 0.3          # (a window of {segment keyframes} + the last frame, 4 iterations x 96 rays per call)
    normal = [c for c in calls if c['idx'] == 0][0]         # first frame: iters_first with colour iterations, all decoder weights move
    assert not normal['refine'] and normal['w_moved'] > 1000 and max(c['w_moved'] for c in last) <= 3 * 96 + 30


def test_get_tensor_from_camera_matches_reference_golden():
    """common.get_tensor_from_camera (src/common.py:354-379) - the product function, not the oracle's - against the
    reference's outputs captured in tests/golden/g3_pose.npz, and its round trip through get_camera_from_tensor."""
    from loopy_slam_amd import common
    from util import load
    g = load('g3_pose')                   # cams -> c2w by the reference's get_camera_from_tensor, back = its get_tensor_from_camera(c2w)
    c2w, ref = torch.from_numpy(g['c2w']), g['back']
    for k in range(c2w.shape[0]):
        m = torch.eye(4)
        m[:3] = c2w[k]
        cam = common.get_tensor_from_camera(m)
        assert cam.dtype == torch.float32 and cam.shape == (7,)
        np.testing.assert_allclose(cam.numpy(), ref[k], rtol=0, atol=1e-6)
        np.testing.assert_allclose(common.get_camera_from_tensor(cam).numpy(), m[:3].numpy(), rtol=0, atol=2e-6)
        tq = common.get_tensor_from_camera(m, Tquad=True)
        assert torch.equal(tq[:3], cam[4:]) and torch.equal(tq[3:], cam[:4])


@pytest.mark.parametrize('backend', backends())
def test_load_reference_shaped_checkpoint(backend, tmp_path):
    """A checkpoint as the REFERENCE writes it mid-run (src/utils/Logger.py:20-65): keyframes carry `dynamic_r_query` (a radius,
    float64) instead of our squared float32 map, and there are no feature tables - load() converts the former and warns about
    the latter."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['use_dynamic_radius'] = True
    cfg['pointcloud'].update(radius_add_max=0.08, radius_add_min=0.02, radius_query_ratio=2, color_grad_threshold=0.15)
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    ps.run(n_frames=3)
    lg = slam.Logger(cfg, None, ps.mapper, ckptsdir=str(tmp_path))
    path = lg.log(2, ps.mapper.keyframe_dict, ps.mapper.keyframe_list, npc=ps.npc, last_log=False)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    for kf in ck['keyframe_dict']:
        kf['dynamic_r_query'] = torch.sqrt(kf.pop('r2_query').double())
    torch.save(ck, path, _use_new_zipfile_serialization=False)
    ps2 = slam.Point_SLAM(cfg, None, eng=eng)
    with pytest.warns(UserWarning, match='no geo_feats'):
        assert slam.Logger.load(path, ps2) == 2
    for a, b in zip(ps2.mapper.keyframe_dict, ps.mapper.keyframe_dict):
        np.testing.assert_allclose(a['r2_query'].cpu().numpy(), b['r2_query'].cpu().numpy(), rtol=2e-7)
    _, color, depth, c2w = ps.frame_reader[3]
    ps2.tracker.track_frame(3, color, depth, c2w)
    ps2.mapper.map_frame(3, color, depth, c2w)                    # would raise KeyError: 'r2_query' without the conversion


@pytest.mark.parametrize('backend', backends())
def test_keyframe_overlap_matches_oracle(backend):
    """Mapper.keyframe_selection_overlap (f3): the batched device projection with the closed-form rigid inverse gives the overlap
    fractions of the reference's per-keyframe numpy loop (oracle.keyframe_overlap_fractions, Mapper.py:250-270) - at most the few
    points that sit on a threshold may flip - and the same ranking; _inv_pose equals the general float64 inverse."""
    from loopy_slam_amd import synthetic as syn
    S = slam
    eng = make_engine(backend)
    cfg = mini_cfg()
    mp = slam.Point_SLAM(cfg, None, eng=eng).mapper
    dev = eng.device
    intr = syn.TUM_INTR                                            # the 20-pixel edge needs a real image size
    mp.H, mp.W, mp.fx, mp.fy, mp.cx, mp.cy = (intr[k] for k in ('H', 'W', 'fx', 'fy', 'cx', 'cy'))
    poses = [syn.loop_pose(i, 40, 'cpu') for i in (0, 3, 7, 12, 20, 33)]
    poses.append(syn.loop_pose(20, 40, 'cpu') @ torch.diag(torch.tensor([1., 1., -1., 1.])))       # a camera looking the other way
    for c in poses:
        ref = torch.linalg.inv(c.double())
        assert torch.allclose(S._inv_pose(c, dev).cpu().double(), ref, atol=2e-6)
    depth, color, c2w = syn.render_frame(5, intr=intr, device='cpu', holes=0.02, n_poses=40)
    g = torch.Generator().manual_seed(3)
    flat = torch.randint(0, depth.numel(), (300,), generator=g)
    Wd = depth.shape[1]
    ro, rd = H.rays_from_uv((flat % Wd).float(), (flat // Wd).float(), c2w, mp.fx, mp.fy, mp.cx, mp.cy)
    gd = depth.reshape(-1)[flat]
    keep = gd > 0
    ro, rd, gd = ro[keep], rd[keep], gd[keep]
    t = torch.linspace(0., 1., 8)
    z = gd[:, None] * 0.8 * (1 - t) + (gd[:, None] + 0.5) * t
    pts = (ro[:, None, :] + rd[:, None, :] * z[..., None]).reshape(-1, 3)
    want = H.keyframe_overlap_fractions(pts.numpy(), [p.numpy() for p in poses], mp.fx, mp.fy, mp.cx, mp.cy, mp.H, mp.W)
    got = mp.overlap_fractions(pts.to(dev), poses).cpu().numpy()
    assert np.abs(got - want).max() <= 3.0 / pts.shape[0], (got, want)
    assert want.max() > 0.5 and want.min() == 0.0                # the case covers seen and unseen keyframes
    sel = mp.keyframe_selection_overlap(color.to(dev), depth.to(dev), c2w.to(dev), [{'est_c2w': p.to(dev)} for p in poses], 3)
    assert len(sel) == 3 and all(want[i] > 0 for i in sel)


@pytest.mark.parametrize('backend', backends())
def test_render_with_sample_near_pcl_matches_oracle(backend):
    """rendering.sample_near_pcl (a21, Renderer.py:152-160, 194-198): rays without a depth reading take their sample depths from the
    probe of the cloud along the ray (LK_FLAG_Z_GIVEN), keep their rendered depth, and the ones that find no cloud are not valid;
    forward and gradients against the oracle on a batch that mixes rays with and without a reading."""
    eng = make_engine(backend)
    cfg = mini_cfg()
    cfg['rendering']['sample_near_pcl'] = True
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    idx, color, depth, c2w = ps.frame_reader[0]
    ps.tracker.track_frame(0, color, depth, c2w)
    ps.mapper.map_frame(0, color, depth, c2w, cur_c2w=c2w)
    npc, dec = ps.npc, ps.shared_decoders
    assert ps.renderer.sample_near_pcl
    g = torch.Generator().manual_seed(1)
    ii = torch.randint(2, 30, (48,), generator=g).float()
    jj = torch.randint(2, 22, (48,), generator=g).float()
    ro, rd = H.rays_from_uv(ii, jj, c2w.cpu(), 26.0, 26.0, 15.5, 11.5)
    gd = depth.cpu()[jj.long(), ii.long()].clone()
    gd[::3] = 0.0                                           # a third of the rays has no reading
    rd[0] = -rd[0]                                          # ... and this one looks away from the cloud: not near it
    geo = npc.get_geo_feats().clone().requires_grad_(True)
    ps.renderer.sigmoid_coefficient = 0.1
    d, u, c, valid = ps.renderer.render_batch_ray(npc, dec, eng.f32(rd), eng.f32(ro), eng.device, 'color', gt_depth=eng.f32(gd),
                                                   npc_geo_feats=geo)
    ((d * 1.3).sum() + (c * c).sum()).backward()
    W = {k: v for k, v in dec.dec.unpack().items()}
    go = npc.get_geo_feats().cpu().clone().requires_grad_(True)
    ocfg = H.RenderCfg(radius_query=cfg['pointcloud']['radius_query'], rel_pos=cfg['model']['encode_rel_pos_in_col'])
    out = H.render_batch(ocfg, ro, rd, gd, npc.cloud_pos().cpu(), go, npc.get_col_feats().cpu(), W, 'color', near_pcl=True)
    ((out['depth'] * 1.3).sum() + (out['color'] * out['color']).sum()).backward()
    assert np.array_equal(valid.cpu().numpy(), out['valid_ray'].numpy()) and not bool(valid[0]) and bool(valid[3::3].any())
    zero = gd <= 0
    assert float(out['depth'].detach()[zero].abs().max()) > 0.1     # the depth of the rays without a reading is rendered, not zeroed
    np.testing.assert_allclose(d.detach().cpu().numpy(), out['depth'].detach().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(c.detach().cpu().numpy(), out['color'].detach().numpy(), rtol=1e-4, atol=2e-5)
    assert float((geo.grad.cpu() - go.grad).abs().max()) < 2e-4 * float(go.grad.abs().max())
