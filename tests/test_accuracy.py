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


def test_welch_interval_against_scipy():
    """tests/welch.py reproduces scipy's Welch test (statistic, degrees of freedom -> p-value) and its half-width is the interval that p-value
    belongs to; the 'resolvable' difference shrinks as 1 / sqrt(n)."""
    from scipy import stats
    import welch as WL
    g = np.random.default_rng(5)
    a, b = 0.35 + 0.05 * g.standard_normal(20), 0.33 + 0.08 * g.standard_normal(6)
    w = WL.welch(a, b)
    r = stats.ttest_ind(a, b, equal_var=False)
    assert abs(w['p_value'] - r.pvalue) <= 1e-12 and abs(w['diff'] / w['se'] - r.statistic) <= 1e-12
    assert abs(w['half_width'] - stats.t.ppf(0.975, w['dof']) * w['se']) <= 1e-15
    # the interval contains 0 exactly when p >= alpha
    assert (abs(w['diff']) <= w['half_width']) == (w['p_value'] >= 0.05)
    big = WL.welch(0.35 + 0.05 * g.standard_normal(80), 0.33 + 0.08 * g.standard_normal(24))
    assert 0.35 * w['resolvable_rel'] < big['resolvable_rel'] < 0.75 * w['resolvable_rel']
    ok, rec = WL.indistinguishable(a, b)
    assert ok and rec['half_width_test'] > rec['half_width']
    ok2, _ = WL.indistinguishable(a + 0.5, b)
    assert not ok2


N_PRODUCT_RUNS = 20          # per config (round-5 review: >= 20, so that the interval is set by the data and not by three runs)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ('room', 'roomfull', 'tum', 'scannet'))
def test_accuracy_matches_the_oracle_loop(name):
    """The product on the GPU (N_PRODUCT_RUNS seeds) against the committed oracle runs of the same config (3-6 runs of 8-60 CPU minutes each):
      (a) ATE RMSE and rendered-depth L1: the difference of the MEANS lies inside its own Welch interval (tests/welch.py; tested at
          alpha = 0.002, the 95 % half-width relative to the oracle's mean is recorded as `resolvable_rel` - the smallest relative difference
          these runs can tell from zero.  It is 5-30 %, not 1 %: single runs of EITHER pipeline scatter by 10-30 %, the trajectories being
          chaotic - README / DESIGN quote that number, not "within 1 %");
      (b) sanity: the product's median run beats the constant-speed prior and dead reckoning by the margins below;
      (c) the numbers go to gpurun_out/accuracy_<name>.json (tools/accuracy_summary.py -> profiles/<tag>_accuracy.json).
    Metric definitions: /root/reference/src/tools/eval_ate.py:195-234 (Horn-aligned translational RMSE), src/Mapper.py:1146-1182 (depth L1)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import accuracy_run as AR
    import welch as WL
    fx = _fixtures(name)
    assert len(fx) >= 3, 'oracle fixtures missing (tools/accuracy_run.py --pipeline oracle)'
    c = fx[0]['config']
    res = []
    for seed in range(c['seed'], c['seed'] + N_PRODUCT_RUNS):
        cfg = AR.make_cfg(os.path.join(ROOT, c['file']), c['frames'], c['rays_per_iteration'], c['iters_scale'], None, seed, int(c['color_refine']), c['scene'])
        res.append(AR.run_product(cfg))
    ha, hl = np.array([r['ate_rmse_cm'] for r in res]), np.array([r['depth_l1_cm'] for r in res])
    oa, ol = np.array([o['ate_rmse_cm'] for o in fx]), np.array([o['depth_l1_cm'] for o in fx])
    ok_a, rec_a = WL.indistinguishable(ha, oa)
    ok_l, rec_l = WL.indistinguishable(hl, ol)
    prior = fx[0]['prior_only']
    summary = dict(config=c, prior_only=prior,
                   runs=[dict(seed=c['seed'] + k, hip_ate=r['ate_rmse_cm'], hip_l1=r['depth_l1_cm'], hip_rot=r['rot_err_deg'], hip_wall_s=r['wall_s']) for k, r in enumerate(res)],
                   oracle_runs=[dict(seed=o['config']['seed'], ate=o['ate_rmse_cm'], l1=o['depth_l1_cm'], rot=o['rot_err_deg'], wall_s=o['wall_s']) for o in fx],
                   ate_rmse_cm=dict(rec_a, a='product', b='oracle'), depth_l1_cm=dict(rec_l, a='product', b='oracle'),
                   statement=f"ATE RMSE: product {rec_a['mean_a']:.3f} cm vs oracle {rec_a['mean_b']:.3f} cm ({100 * rec_a['rel_diff']:+.1f} %), indistinguishable at "
                             f"+-{100 * rec_a['resolvable_rel']:.0f} % (95 % Welch, n = {rec_a['n_a']} / {rec_a['n_b']}); depth L1: {rec_l['mean_a']:.4f} vs "
                             f"{rec_l['mean_b']:.4f} cm ({100 * rec_l['rel_diff']:+.1f} %), +-{100 * rec_l['resolvable_rel']:.0f} %")
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, f'accuracy_{name}.json'), 'w') as f:
            json.dump(summary, f, indent=1)
    print(summary['statement'])
    assert ok_a, ('ATE RMSE', rec_a)
    assert ok_l, ('depth L1', rec_l)
    # the tracker does its job: the median run beats the one-step constant-speed prior (config 1: by any margin; the 2 000-ray configs lose
    # more frames) and dead reckoning by 1.5 x
    assert float(np.median(ha)) < (prior['one_step_ate_cm'] if name in ('room', 'roomfull') else prior['dead_reckoning_ate_cm'])
    assert float(np.median(ha)) < prior['dead_reckoning_ate_cm'] / 1.5
