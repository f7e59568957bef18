"""TEACHER FORCING: the oracle's SLAM loop (tests/oracle_slam.py: the reference's Tracker.run / Mapper.run bodies on the pinned oracle) runs
LIVE, and every tracking call and every optimize_map call it makes is replayed through the product - lk_track_frame / lk_map_frame, the
native loops - from the oracle's own state at that call (cloud after the frame's insertion, feature tables, decoder weights, exposure
state, keyframe window, frustum rows) and the oracle's own draws.

tests/test_accuracy.py compares two chaotic trajectories through their statistics (ATE means over seeds); here the chaos is cut at every
call, so each call is a deterministic comparison: the per-iteration losses, the chosen pose, the stepped rows and decoders.  A 1 % error
in a loop that the ATE band cannot see shows here as a 1 % loss difference in the first iterations of the first frame.

  config 1   configs/Synthetic/room.yaml at 500 rays per iteration (BASELINE config 1), 10 frames of the hand-held walk through the
             furnished room: frames 2..9 tracked (40 iterations), frames 0 and 5 mapped (iters_first shortened, see CFG1)
  TUM        configs/TUM_RGBD/freiburg1_desk.yaml: 5 000-ray tracking from the gradient-pixel pool, 10 000-ray mapping over the keyframe
             window with per-pixel dynamic radii - 10 iterations per call
  ScanNet    configs/ScanNet/scene0000.yaml: the same with exposure encoding (per-sample affine in the tracker, per-keyframe affine on the
             rendered logits in the mapper, mlp_exposure and the frame's feature stepped), surface ratios 0.96 / 1.04, cropped frames

Reference: src/Tracker.py:281-409 (102-197), src/Mapper.py:347-807, 835-1049.  Measured values: gpurun_out/teacher_forced.json."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import oracle_slam as OS
from oracle import hotpath as H
from loopy_slam_amd import config, core, slam, steps
from test_loops_at_size import param_error_stats
from util import make_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REPORT = {}


def _record(case, **kv):
    _REPORT.setdefault(case, {}).update(kv)
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'teacher_forced.json'), 'w') as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True, default=float)


def _sync(eng):
    if eng.device.type == 'cuda':
        torch.cuda.synchronize()


def _exposure_module(W, dev):
    m = torch.nn.Sequential(torch.nn.Linear(8, 128), torch.nn.Softplus(beta=100), torch.nn.Linear(128, 12))
    with torch.no_grad():
        m[0].weight.copy_(W['color_decoder.mlp_exposure.linear1.weight']); m[0].bias.copy_(W['color_decoder.mlp_exposure.linear1.bias'])
        m[2].weight.copy_(W['color_decoder.mlp_exposure.linear2.weight']); m[2].bias.copy_(W['color_decoder.mlp_exposure.linear2.bias'])
    return m.to(dev)


class Replayer:
    """Replays the oracle's calls through steps.TrackOptimizer / steps.MapOptimizer (the classes slam.Tracker / slam.Mapper drive)."""

    def __init__(self, eng, cfg, case, tol):
        self.eng, self.cfg, self.case, self.tol = eng, cfg, case, tol
        c = cfg['cam']
        e = c.get('crop_edge', 0) or 0
        self.H, self.W = c['H'] - 2 * e, c['W'] - 2 * e
        self.intr = (c['fx'], c['fy'], c['cx'] - e, c['cy'] - e)
        self.track_stats, self.map_stats = [], []
        self._map = None            # (N, knn, dpos): the index of the current cloud, rebuilt when the oracle's map has grown

    def _index(self, pos):
        if self._map is None or self._map[0] != pos.shape[0]:
            dpos = self.eng.f32(pos)
            knn = core.KnnIndex(self.eng, capacity=pos.shape[0], cell_size=max(self.cfg['pointcloud']['radius_query'], 1e-3))     # as slam.NeuralPointCloud
            knn.build(dpos)
            self._map = (pos.shape[0], knn, dpos)
        return self._map[1], self._map[2]

    # ---- one tracking call (Tracker.py:313-401)
    def on_track(self, rec, out):
        eng, cfg, t = self.eng, self.cfg, self.cfg['tracking']
        rcfg = slam.render_cfg_from(cfg, cfg['rendering']['sigmoid_coef_tracker'])
        dec = core.DecoderBlob(eng).pack(rec['W'])
        knn, dpos = self._index(rec['pos'])
        dgeo, dcol = eng.f32(rec['geo']), eng.f32(rec['col'])
        flat = rec['flat']
        to = steps.TrackOptimizer(eng, rcfg, dec, knn, dpos, dgeo, dcol, flat.shape[1], rec['lr'], separate_lr=rec['separate'],
                                  w_color=t['w_color_loss'], use_color=t['use_color_in_tracking'], dynamic_radius=rec['r2_map'] is not None,
                                  handle_dynamic=t.get('handle_dynamic', True))
        assert to.native_loop
        xp = feat = None
        if rec['xfeat'] is not None:
            mlp = _exposure_module(rec['W'], eng.device)
            feat = eng.f32(rec['xfeat']).clone().requires_grad_(True)
            xp = (mlp, feat)
        best, log = to.track(eng.f32(rec['cam']), eng.f32(rec['depth']), eng.f32(rec['color']), rec['iters'], rec['win'], self.intr,
                             flat.to(torch.int32).to(eng.device), r2_map=eng.f32(rec['r2_map']) if rec['r2_map'] is not None else None, exposure=xp)
        _sync(eng)
        kl, ol = log[:, 0].cpu().numpy().astype(np.float64), np.array(out['losses'])
        km, om = log[:, 3].cpu().numpy().astype(int), np.array(out['masked'])
        rel = np.abs(kl - ol) / np.abs(ol)
        s = dict(frame=len(self.track_stats), loss_rel_first=float(rel[0]), loss_rel_first10=float(rel[:10].max()), loss_rel_all=float(rel.max()),
                 loss_rel_last=float(rel[-1]), masked_diff=int(np.abs(km - om).max()), pose_err=float((best.cpu() - out['best']).abs().max()),
                 chosen=(int(np.argmin(kl)), int(np.argmin(ol))), moved=float((out['best'] - rec['cam']).abs().max()), points=int(rec['pos'].shape[0]))
        if feat is not None:
            s['xfeat_err'] = float((feat.detach().cpu() - out['xfeat']).abs().max())
            s['xfeat_moved'] = float((out['xfeat'] - rec['xfeat']).abs().max())
            s['xb2_err'] = float((mlp[2].bias.detach().cpu() - out['W_exposure']['color_decoder.mlp_exposure.linear2.bias']).abs().max())
        self.track_stats.append(s)
        _record(self.case, track=self.track_stats)
        tl = self.tol
        assert np.isfinite(kl).all()
        assert s['loss_rel_first'] <= tl['loss_first'] and s['loss_rel_first10'] <= tl['loss_first10'] and s['loss_rel_all'] <= tl['loss_all'], s
        assert s['masked_diff'] <= tl['masked'], s
        assert s['pose_err'] <= tl['pose'], s
        if feat is not None:
            assert s['xfeat_err'] <= 2e-4 and s['xb2_err'] <= 3e-4, s

    # ---- one optimize_map call (Mapper.py:562-735)
    def on_map(self, rec, out):
        eng, cfg = self.eng, self.cfg
        rcfg = slam.render_cfg_from(cfg, cfg['rendering']['sigmoid_coef_mapper'])
        W = rec['W']
        dec = core.DecoderBlob(eng).pack(W)
        knn, dpos = self._index(rec['pos'])
        dgeo, dcol = eng.f32(rec['geo']).clone(), eng.f32(rec['col']).clone()
        N, R, iters = rec['pos'].shape[0], rec['R'], rec['iters']
        xp = feats = None
        if rec['xfeats'] is not None:
            mlp = _exposure_module(W, eng.device)
            feats = [eng.f32(x).clone().requires_grad_(True) for x in rec['xfeats']]
            xp = (mlp, feats)
        mo = steps.MapOptimizer(eng, rcfg, dec, knn, dpos, dgeo, dcol, None, R, rec['lrs'], w_color=rec['w_color'], dynamic_radius=rec['rstack'] is not None,
                                fix_color_decoder=rec['fix_color_decoder'], exposure=xp)
        assert mo._takes_native_loop()
        rows = rec['rows']
        all_rows = rows.numel() == N and bool((rows == torch.arange(N)).all())
        if all_rows:
            mo.begin_frame()
        else:
            mask = torch.zeros(N, dtype=torch.uint8)
            mask[rows] = 1
            mo.new_frame(rows.to(torch.int32).to(eng.device), mask.to(eng.device))
        log = eng.zeros(iters, 4)
        frames = (eng.f32(rec['dstack']), eng.f32(rec['cstack']), eng.f32(rec['pstack']), eng.f32(rec['rstack']) if rec['rstack'] is not None else None)
        n_geo = min(iters, rec['geo_iters'] + 1)                    # stage 'geometry' while it <= geo_iters (Mapper.py:594-597)
        mo.run(iters, n_geo, frames, rec['rnd'].to(torch.int32).to(eng.device), rec['fid'].to(torch.int32).to(eng.device), (0, self.H, 0, self.W),
               self.intr, self.H, self.W, log)
        mo.finish()
        _sync(eng)
        kl, ol = log[:, 0].cpu().numpy().astype(np.float64), np.array(out['losses'])
        rel = np.abs(kl - ol) / np.abs(ol)
        s = dict(idx=rec['idx'], iters=iters, n_geo=n_geo, rays=R, rows=int(rows.numel()), points=N, frames=rec['F'], loss_rel_first=float(rel[0]),
                 loss_rel_first10=float(rel[:10].max()), loss_rel_all=float(rel.max()), loss_rel_last=float(rel[-1]), loss_first=float(ol[0]), loss_last=float(ol[-1]))
        lr_g, lr_c = rec['lrs']['geometry'][1], rec['lrs']['color'][2]
        gk, ck = dgeo.cpu(), dcol.cpu()
        for name, mine, ref, before in (('geo', gk[rows], out['geo_rows'], rec['geo'][rows]), ('col', ck[rows], out['col_rows'], rec['col'][rows])):
            s.update({f'{name}_{k}': v for k, v in param_error_stats(mine, ref, before).items()})
        other = torch.ones(N, dtype=torch.bool)
        other[rows] = False
        s['untouched_rows_equal'] = bool(torch.equal(gk[other], rec['geo'][other]) and torch.equal(ck[other], rec['col'][other]))
        Wk = dec.unpack()
        worst = 0.0
        for n in rec['dec_names']:
            if n in Wk:
                st = param_error_stats(Wk[n].reshape(out['W'][n].shape), out['W'][n], W[n])
                worst = max(worst, (st['err_q999'] - 0.1 * st['moved_max']) / max(1.0, float(W[n].abs().max())))
        s['decoder_excess_q999'] = worst
        if feats is not None:
            s['xb2_err'] = float((mlp[2].bias.detach().cpu() - out['W']['color_decoder.mlp_exposure.linear2.bias']).abs().max()) \
                if 'color_decoder.mlp_exposure.linear2.bias' in out['W'] else 0.0
            s['xfeat_err'] = float((feats[-1].detach().cpu() - out['xfeat']).abs().max())
            s['xfeat_moved'] = float((out['xfeat'] - rec['xfeats'][-1]).abs().max())
            s['keyframe_feats_constant'] = all(torch.equal(f.detach().cpu(), x) for f, x in zip(feats[:-1], rec['xfeats'][:-1]))
        self.map_stats.append(s)
        _record(self.case, map=self.map_stats)
        tl = self.tol
        assert np.isfinite(kl).all()
        assert s['loss_rel_first'] <= tl['map_loss_first'] and s['loss_rel_first10'] <= tl['map_loss_first10'] and s['loss_rel_all'] <= tl['map_loss_all'], s
        assert s['untouched_rows_equal'], s
        n_col = max(1, iters - n_geo)
        assert s['geo_err_q99'] <= 0.02 * lr_g * iters ** 0.5 and s['geo_err_max'] <= 2.0 * lr_g * iters, s
        assert s['col_err_q99'] <= 0.02 * lr_c * n_col ** 0.5 and s['col_err_max'] <= 2.0 * lr_c * n_col, s
        assert s['decoder_excess_q999'] <= 2e-4, s
        if feats is not None:
            assert s['xfeat_err'] <= 3e-4 and s['xb2_err'] <= 3e-4 and s['keyframe_feats_constant'], s


def _load(path, **over):
    cfg = copy.deepcopy(config.load_config(os.path.join(ROOT, path), os.path.join(ROOT, 'configs/point_slam.yaml')))
    for sec, kv in over.items():
        cfg[sec].update(kv)
    return cfg


# BASELINE config 1: 500 rays per iteration, the room config's iteration counts - with the first frame's 1 500 iterations cut to 300 (an
# oracle iteration at 500 rays is 0.03-0.08 s: 1 500 of them would be two minutes of every run for the same code path)
CFG1 = dict(tracking=dict(pixels=500), mapping=dict(pixels=500, iters_first=300, geo_iter_first=120, color_refine=False),
            data=dict(n_frames=10, motion='handheld', scene='furnished'))
TOL1 = dict(loss_first=2e-4, loss_first10=5e-4, loss_all=5e-3, masked=1, pose=1e-4, map_loss_first=2e-4, map_loss_first10=1e-3, map_loss_all=2e-2)
# TUM / ScanNet at their own ray budgets, 10 iterations per call, three frames (two tracked, frames 0 and 2 mapped)
CFG_TUM = dict(tracking=dict(iters=10), mapping=dict(iters_first=10, geo_iter_first=3, iters=10, every_frame=2, keyframe_every=1, color_refine=False,
                                                       min_iter_ratio=1.0),
               data=dict(n_frames=4, motion='handheld', scene='furnished'))
TOL_X = dict(loss_first=2e-4, loss_first10=5e-4, loss_all=5e-4, masked=1, pose=1e-4, map_loss_first=2e-4, map_loss_first10=1e-3, map_loss_all=1e-3)


def run_teacher_forced(eng, case, cfg, n_frames, tol):
    reader = slam.SyntheticRoomDataset(cfg, 'cpu', n_frames)            # (the product's frame reader: the synthetic sequence, cropped as the config says)
    frames = [reader[i] for i in range(n_frames)]
    o = OS.OracleSLAM(cfg, frames)
    rp = Replayer(eng, cfg, case, tol)
    o.on_track, o.on_map = rp.on_track, rp.on_map
    o.run(n_frames)
    return o, rp


@pytest.mark.gpu
def test_config1_ten_frames_teacher_forced():
    cfg = _load('configs/Synthetic/room.yaml', **CFG1)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    o, rp = run_teacher_forced(make_engine('hip'), 'config1-500rays-10frames', cfg, 10, TOL1)
    assert len(rp.track_stats) == 8 and len(rp.map_stats) == 3          # frames 2..9 tracked; frames 0, 5 and the last one (9) mapped
    assert max(s['pose_err'] for s in rp.track_stats) <= 1e-4
    # the oracle itself tracked: better than the constant-speed prior it starts every frame from
    gt = torch.stack([f[3] for f in o.frames[:10]])
    assert OS.ate_rmse(o.est[:10], gt) < OS.prior_only_metrics(gt)['one_step_ate_cm'] / 100


@pytest.mark.gpu
@pytest.mark.parametrize('name,path', (('tum', 'configs/TUM_RGBD/freiburg1_desk.yaml'), ('scannet', 'configs/ScanNet/scene0000.yaml')))
def test_tum_scannet_budgets_teacher_forced(name, path):
    cfg = _load(path, **CFG_TUM)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    o, rp = run_teacher_forced(make_engine('hip'), f'{name}-fullrays-10it', cfg, 4, TOL_X)
    assert len(rp.track_stats) == 2 and len(rp.map_stats) >= 2
    assert rp.track_stats[0]['points'] > 10_000 and rp.map_stats[-1]['rays'] >= 9_000
    if name == 'scannet':
        assert all('xfeat_err' in s for s in rp.track_stats) and rp.map_stats[-1]['xfeat_moved'] > 1e-4


def test_teacher_forced_miniature_on_the_emulator():
    """The same replay on the host emulator at a miniature size (24 x 32 frames, 48 / 64 rays): the harness itself in the CPU suite."""
    cfg = _load('configs/Synthetic/room.yaml', tracking=dict(ignore_edge_W=2, ignore_edge_H=2, pixels=48, iters=6),
                mapping=dict(pixels=64, pixels_adding=400, iters=6, iters_first=12, geo_iter_first=4, every_frame=2, keyframe_every=2,
                             mapping_window_size=4, color_refine=False),
                pointcloud=dict(radius_add=0.12, radius_query=0.24, radius_min=0.06), data=dict(n_frames=5, motion='handheld', scene='furnished'))
    cfg['cam'].update(H=24, W=32, fx=26.0, fy=26.0, cx=15.5, cy=11.5)
    torch.set_num_threads(4)
    tol = dict(TOL1, loss_all=2e-3, map_loss_all=5e-3)
    o, rp = run_teacher_forced(make_engine('emu'), 'emu-miniature', cfg, 5, tol)
    assert len(rp.track_stats) == 3 and len(rp.map_stats) == 3
    _REPORT.clear()
