"""Trajectory-level accuracy of the drop-in pipeline against the CPU oracle's SLAM loop (tests/oracle_slam.py) - BASELINE config 1
("50 frames, 500 rays/iter") on the synthetic hand-held sequence through the furnished room, with the reference's metrics: ATE RMSE
(src/tools/eval_ate.py:44-79,195-234) and rendered-depth L1 at the estimated poses (src/Mapper.py:1146-1182).

The oracle's runs are fixtures: tests/golden/accuracy_*.json, written by tools/accuracy_run.py --pipeline oracle in the build container
(8-9 minutes of CPU per 50-frame run; the generating command is recorded in each file).  The two pipelines start from the same random-init
decoders and read the same frames; their fp32 trajectories separate within a frame or two (Adam's first steps are sign-like), so what is
compared is the METRICS, seed by seed and in the mean over the seeds."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import oracle_slam as OS
from util import make_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _fixtures(name):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLD, f'accuracy_{name}_oracle_s*.json'))):
        with open(f) as fh:
            out.append(json.load(fh))
    return out


def test_metrics_on_known_trajectories():
    """ate_rmse is invariant under a rigid motion of the estimate and measures a known perturbation; the prior-only baselines are what they
    say (eval_ate.py:44-79)."""
    from loopy_slam_amd import synthetic as syn
    gt = torch.stack([syn.handheld_pose(k) for k in range(30)])
    c, s = np.cos(0.3), np.sin(0.3)
    Rz = torch.tensor([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
    Rz[:3, 3] = torch.tensor([0.5, -0.2, 0.1])
    moved = Rz @ gt
    assert OS.ate_rmse(moved, gt) < 1e-6
    g = torch.Generator().manual_seed(0)
    noisy = gt.clone()
    noisy[:, :3, 3] += 0.01 * torch.randn(30, 3, generator=g)
    a = OS.ate_rmse(noisy, gt)
    assert 0.012 < a < 0.022                     # sigma sqrt(3) = 1.73 cm, minus what the alignment absorbs
    p = OS.prior_only_metrics(gt)
    assert p['dead_reckoning_ate_cm'] > 2 * p['one_step_ate_cm'] > 0.5         # the sequence's velocity changes: a constant-speed guess is off by > 2.5 mm
    step = (gt[1:, :3, 3] - gt[:-1, :3, 3]).norm(dim=1)
    assert float(step.mean()) > 0.01                                             # >= 1 cm per frame


def test_oracle_loop_and_product_agree_on_a_miniature_sequence():
    """Both pipelines on the host (the product on the emulator: device = cpu, so both consume the SAME draws): the map after the first
    frame is the same cloud point for point, the first losses agree to rounding, and four frames later both still track."""
    import copy
    from loopy_slam_amd import slam, config
    torch.set_num_threads(4)
    cfg = copy.deepcopy(config.load_config(os.path.join(ROOT, 'configs/Synthetic/room.yaml'), os.path.join(ROOT, 'configs/point_slam.yaml')))
    cfg['cam'].update(H=24, W=32, fx=26.0, fy=26.0, cx=15.5, cy=11.5)
    cfg['tracking'].update(ignore_edge_W=2, ignore_edge_H=2, pixels=48, iters=6)
    cfg['mapping'].update(pixels=64, pixels_adding=400, iters=6, iters_first=12, geo_iter_first=4, every_frame=2, keyframe_every=2,
                          mapping_window_size=4, color_refine=False)
    cfg['pointcloud'].update(radius_add=0.12, radius_query=0.24, radius_min=0.06)
    cfg['data'].update(n_frames=5, motion='handheld', scene='furnished')
    eng = make_engine('emu')
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    frames = [ps.frame_reader[i] for i in range(5)]
    o = OS.OracleSLAM(cfg, frames)
    idx, c, d, p = frames[0]
    ps.tracker.track_frame(0, c, d, p); o.track_frame(0, c, d, p)
    ps.mapper.map_frame(0, c, d, p, cur_c2w=p); o.map_frame(0, c, d, p, p)
    assert ps.npc.pts_num() == o.pos.shape[0] and torch.equal(ps.npc.cloud_pos(), o.pos)          # the same insertion, bit for bit
    l_p, l_o = float(ps.mapper.last_log[0, 0]), o.map_log[-1]['loss_first']
    assert abs(l_p - l_o) <= 2e-4 * abs(l_o)
    assert o.map_log[-1]['iters'] == ps.mapper.last_num_joint_iters
    for k in range(1, 5):
        idx, c, d, p = frames[k]
        e1 = ps.tracker.track_frame(idx, c, d, p)
        e2 = o.track_frame(idx, c, d, p)
        if idx % 2 == 0:
            ps.mapper.map_frame(idx, c, d, p, cur_c2w=e1); o.map_frame(idx, c, d, p, e2)
            assert ps.npc.pts_num() % 3 == 0 and abs(ps.npc.pts_num() - o.pos.shape[0]) <= 0.05 * o.pos.shape[0]
    est_p, est_o = ps.estimate_c2w_list[:5], o.est[:5]
    gt = torch.stack([f[3] for f in frames])
    assert torch.equal(est_p[:2], gt[:2]) and torch.equal(est_o[:2], gt[:2])                      # the first two frames keep the given pose
    assert float((est_p[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.08 and float((est_o[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()) < 0.08


@pytest.mark.gpu
@pytest.mark.parametrize('name', ('room', 'roomfull', 'tum', 'scannet'))
def test_accuracy_matches_the_oracle_loop(name):
    """The product on the GPU against the committed oracle runs of the same config and seeds:
      (a) rendered-depth L1 within 5 % of the oracle's, seed by seed (it is set by the map, which both build from the same draws' distribution);
      (b) ATE RMSE: every run beats the one-step constant-speed prior and is >= 3x better than dead reckoning, and the MEAN over the seeds is
          within 5 % of the oracle's mean or within three standard errors of the difference (a two-sigma bound fails one honest run in twenty) (the ATE of one 50-frame run scatters by ~10 %
          from seed to seed in BOTH pipelines - measured, profiles/r4_accuracy.json);
      (c) the numbers go to gpurun_out/accuracy_<name>.json."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import accuracy_run as AR
    fx = _fixtures(name)
    if name != 'room' and not fx:
        pytest.skip(f'no oracle fixture for the {name} config')
    assert len(fx) >= (3 if name == 'room' else 1), 'oracle fixtures missing (tools/accuracy_run.py --pipeline oracle)'
    if name != 'room':
        # TUM / ScanNet configs (dynamic radii, gradient-pool tracking pixels, exposure encoding) at 2 000 rays per iteration - at config 1's
        # 500 rays BOTH pipelines lose track on this sequence (oracle 22 cm, product 14-31 cm): ONE oracle run each (25-60 minutes of CPU),
        # five product runs (three at the room's own budget); a band instead of a statistical bound.  roomfull: the room config at its OWN ray budget (1 500 / 5 000 rays per
        # iteration - the bench workload's), one oracle run of 50 minutes; the depth L1 there is held to 8 % (the runs of either pipeline scatter by +-4 % around 0.047 cm)
        o = dict(fx[0])                       # (more than one oracle run of the config: the band is around their MEDIANS - one oracle run in
        # three of the ScanNet config drifts to 4.4 cm ATE, and its depth L1 with it; the product's eight runs stay at 1.5-2.3 cm)
        o['ate_rmse_cm'] = float(np.median([f['ate_rmse_cm'] for f in fx]))
        # depth L1 of the runs that kept track (ATE within 1.5 x the pipeline's median): a drifted run's depth L1 is up to twice the others'
        held = lambda ate, l1: float(np.median([l for a, l in zip(ate, l1) if a <= 1.5 * np.median(ate)]))
        o['depth_l1_cm'] = held([f['ate_rmse_cm'] for f in fx], [f['depth_l1_cm'] for f in fx])
        c = o['config']
        res = []
        for seed in range(c['seed'], c['seed'] + (3 if name == 'roomfull' else 5)):
            cfg = AR.make_cfg(os.path.join(ROOT, c['file']), c['frames'], c['rays_per_iteration'], c['iters_scale'], None, seed, int(c['color_refine']), c['scene'])
            res.append(AR.run_product(cfg))
        ha = np.array([r['ate_rmse_cm'] for r in res]); hl = np.array([r['depth_l1_cm'] for r in res])
        out = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(out):
            with open(os.path.join(out, f'accuracy_{name}.json'), 'w') as f:
                json.dump(dict(config=c, oracle=dict(ate_rmse_cm=o['ate_rmse_cm'], depth_l1_cm=o['depth_l1_cm'], rot_err_deg=o['rot_err_deg']),
                               hip_ate_rmse_cm=ha.tolist(), hip_depth_l1_cm=hl.tolist()), f, indent=1)
        prior = o['prior_only']
        assert 0.4 * o['ate_rmse_cm'] <= float(np.median(ha)) <= 2.5 * o['ate_rmse_cm'], (ha.tolist(), o['ate_rmse_cm'])
        # (medians on both sides: a run of either pipeline that drifts - one in three to five does on the ScanNet config - takes its depth L1 with it)
        assert abs(held(ha, hl) / o['depth_l1_cm'] - 1) <= (0.08 if name == 'roomfull' else 0.2), (hl.tolist(), o['depth_l1_cm'])
        assert float(np.median(ha)) < prior['dead_reckoning_ate_cm'] / 1.5
        return
    rows = []
    for o in fx:
        c = o['config']
        cfg = AR.make_cfg(os.path.join(ROOT, c['file']), c['frames'], c['rays_per_iteration'], c['iters_scale'], None, c['seed'], int(c['color_refine']), c['scene'])
        res = AR.run_product(cfg)
        rows.append(dict(seed=c['seed'], hip_ate=res['ate_rmse_cm'], oracle_ate=o['ate_rmse_cm'], hip_l1=res['depth_l1_cm'], oracle_l1=o['depth_l1_cm'],
                         hip_rot=res['rot_err_deg'], oracle_rot=o['rot_err_deg'], prior=o['prior_only'], hip_wall_s=res['wall_s'], oracle_wall_s=o['wall_s']))
    ha, oa = np.array([r['hip_ate'] for r in rows]), np.array([r['oracle_ate'] for r in rows])
    hl, ol = np.array([r['hip_l1'] for r in rows]), np.array([r['oracle_l1'] for r in rows])
    n = len(rows)
    se = float(np.sqrt(ha.var(ddof=1) / n + oa.var(ddof=1) / n))
    summary = dict(config=fx[0]['config'], runs=rows, ate_mean_cm=dict(hip=float(ha.mean()), oracle=float(oa.mean())),
                   ate_sd_cm=dict(hip=float(ha.std(ddof=1)), oracle=float(oa.std(ddof=1))), ate_mean_rel_diff=float(ha.mean() / oa.mean() - 1),
                   ate_diff_standard_error_cm=se, depth_l1_mean_cm=dict(hip=float(hl.mean()), oracle=float(ol.mean())),
                   depth_l1_max_rel_diff=float(np.abs(hl / ol - 1).max()))
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, f'accuracy_{name}.json'), 'w') as f:
            json.dump(summary, f, indent=1)
    prior = rows[0]['prior']
    assert float(np.abs(hl / ol - 1).max()) <= 0.05, summary['depth_l1_max_rel_diff']
    assert abs(hl.mean() / ol.mean() - 1) <= 0.03
    assert ha.max() < prior['one_step_ate_cm'] and 3.0 * ha.max() <= prior['dead_reckoning_ate_cm']
    assert oa.max() < prior['one_step_ate_cm'] and 3.0 * oa.max() <= prior['dead_reckoning_ate_cm']
    assert abs(ha.mean() - oa.mean()) <= max(0.05 * oa.mean(), 3.0 * se), (float(ha.mean()), float(oa.mean()), se)
