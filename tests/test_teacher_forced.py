"""TEACHER FORCING: the oracle's SLAM loop (tests/oracle_slam.py: the reference's Tracker.run / Mapper.run bodies on the pinned oracle) runs
LIVE, and every tracking call and every optimize_map call it makes is replayed through the product - lk_track_frame / lk_map_frame, the
native loops - from the oracle's own state at that call (cloud after the frame's insertion, feature tables, decoder weights, exposure
state, keyframe window, frustum rows) and the oracle's own draws.

tests/test_accuracy.py compares two chaotic trajectories through their statistics (ATE means over seeds); here the chaos is cut at every
call.  Inside a call it is still there - measured: the oracle's own 300-iteration mapping call of frame 0, re-run with its feature table
perturbed by 1e-7 relative, ends 4.7 % apart in the loss (11 % at the worst iteration) and 0.2 apart in 1 % of the stepped entries; a
40-iteration tracking call at the configured rate separates by a factor of ~10 per iteration from the sixth iteration on - so every call is
compared (a) TIGHTLY where rounding has not been amplified yet: the first iterations (loss 5e-5 / 1e-4, masked-ray counts, candidate
poses) and, for tracking calls, the whole call once more with the rate divided by 200 ("stiff": every iteration of the launch sequence on
all but identical inputs); (b) over the whole call against a YARDSTICK: the oracle loop re-run with the feature tables perturbed by 1e-7 -
the product may be 3 x as far from the oracle as that.  A 1 % error in a loop, which the ATE band cannot see, fails (a) in the first
iteration of the first frame.

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
import test_loops_at_size as L
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

    def __init__(self, eng, cfg, case):
        self.eng, self.cfg, self.case = eng, cfg, case
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
        dflat = flat.to(torch.int32).to(eng.device)
        dr2 = eng.f32(rec['r2_map']) if rec['r2_map'] is not None else None
        ddepth, dcolor = eng.f32(rec['depth']), eng.f32(rec['color'])
        c2w = lambda c: H.quat_to_c2w(c.double())           # poses as [R | t]: |q| is a free direction (tests/test_loops_at_size.py)

        def product(lr, exposure):
            to = steps.TrackOptimizer(eng, rcfg, dec, knn, dpos, dgeo, dcol, flat.shape[1], lr, separate_lr=rec['separate'], w_color=t['w_color_loss'],
                                      use_color=t['use_color_in_tracking'], dynamic_radius=dr2 is not None, handle_dynamic=t.get('handle_dynamic', True))
            assert to.native_loop
            best, log = to.track(eng.f32(rec['cam']), ddepth, dcolor, rec['iters'], rec['win'], self.intr, dflat, r2_map=dr2, exposure=exposure)
            _sync(eng)
            return best.cpu(), log[:, 0].cpu().numpy().astype(np.float64), log[:, 3].cpu().numpy().astype(int), to._keep_native[5].cpu()

        xp = feat = mlp = None
        if rec['xfeat'] is not None:
            mlp = _exposure_module(rec['W'], eng.device)
            feat = eng.f32(rec['xfeat']).clone().requires_grad_(True)
            xp = (mlp, feat)
        best, kl, km, hist = product(rec['lr'], xp)
        ol, om = np.array(out['losses']), np.array(out['masked'])
        rel = np.abs(kl - ol) / np.abs(ol)
        s = dict(frame=len(self.track_stats), iters=rec['iters'], rays=int(flat.shape[1]), loss_rel=rel.tolist(), masked_diff=np.abs(km - om).tolist(),
                 pose_err=float((c2w(best) - c2w(out['best'])).abs().max()), chosen=(int(np.argmin(kl)), int(np.argmin(ol))),
                 best_loss_ratio=float(kl.min() / ol.min()), moved=float((c2w(out['best']) - c2w(rec['cam'])).abs().max()), points=int(rec['pos'].shape[0]))
        if feat is not None:
            s['xfeat_err'] = float((feat.detach().cpu() - out['xfeat']).abs().max())
            s['xfeat_moved'] = float((out['xfeat'] - rec['xfeat']).abs().max())
            s['xb2_err'] = float((mlp[2].bias.detach().cpu() - out['W_exposure']['color_decoder.mlp_exposure.linear2.bias']).abs().max())
        else:
            # the stiff call: the same draws with the rate divided by 200, product against the oracle loop (test_loops_at_size.oracle_track_loop)
            lr_s = rec['lr'] / 200.0
            render = L.TreeRender(rec['pos'], rcfg.rel_pos, near=rcfg.near_surface, far=rcfg.far_surface)
            render.cfg.radius_query, render.cfg.coef, render.cfg.min_nn = rcfg.radius_query, rcfg.coef, rcfg.min_nn
            sl, sm, sc, sb = L.oracle_track_loop(render, rec['geo'], rec['col'], rec['W'], rec['cam'], rec['depth'], rec['color'], flat, rec['win'], self.intr,
                                                 lr_s, rec['separate'], w_color=t['w_color_loss'], r2_map=rec['r2_map'])
            bs, ks, kms, hs = product(lr_s, None)
            s['stiff_loss_rel'] = (np.abs(ks - np.array(sl)) / np.abs(np.array(sl))).tolist()
            s['stiff_masked_diff'] = np.abs(kms - np.array(sm)).tolist()
            s['stiff_pose_err'] = [float((c2w(hs[it]) - c2w(sc[it])).abs().max()) for it in range(rec['iters'])]
            s['stiff_lr'] = lr_s
        self.track_stats.append(s)
        _record(self.case, track=self.track_stats)

    def check_track(self, s, exact_iters=2):
        """Assertions on one tracking call's statistics (after the run: a failing call leaves every call's numbers in the report).
        Measured on the chip (gpurun_out/teacher_forced.json, round 5): first five iterations <= 4e-5 at 500 rays (<= 3e-6 in seven of eight
        calls), stiff calls <= 3.4e-4 at 500 rays / <= 7e-7 at 5 000 (ONE sample whose eighth neighbour sits on the radius edge is 1 / (5 R) of the
        batch), stiff poses <= 2.1e-6 = 0.2 steps after 40 iterations; at the configured rate the calls end up to 11 % apart in the loss and up
        to 0.016 in the pose - as far as the oracle is from itself after a 1e-7 perturbation (see the module header)."""
        n = min(exact_iters, s['iters'])
        rel, dm = np.array(s['loss_rel']), np.array(s['masked_diff'])
        # (ONE sample whose eighth neighbour sits on the radius edge, or one ray on the loss mask's threshold, is 1 / (5 R) ... 1 / R of the batch:
        # 1e-4 at 5 000 rays, 5e-4 at 500)
        assert np.isfinite(rel).all() and L.stiff_call_ok(rel[:n], dm[:n], s['rays'], max(1e-4, 0.25 / s['rays'])), ('first iterations', s)
        assert dm.max() <= max(5, s['rays'] // 100), ('masked rays', s)
        lr = self.cfg['tracking']['lr']
        # the whole call at the configured rate: a sanity band only (chaotic from ~ the sixth iteration on)
        assert rel.max() <= 0.3 and abs(s['best_loss_ratio'] - 1) <= 0.1 and s['pose_err'] <= lr * s['iters'], ('whole call', s)
        if 'stiff_loss_rel' in s:
            sr, sp = np.array(s['stiff_loss_rel']), np.array(s['stiff_pose_err'])
            # (one ray's worth of loss: measured <= 6.3e-4 at 500 rays - 32 stiff calls on four boxes - and <= 7e-7 at 5 000; a ray on the loss
            # mask's threshold: test_loops_at_size.stiff_call_ok)
            assert L.stiff_call_ok(sr, s['stiff_masked_diff'], s['rays'], max(5e-5, 1.0 / s['rays'])), ('stiff call', s)
            assert (sp <= 0.1 * s['stiff_lr'] * (1 + np.arange(sp.size)) + 2e-7).all(), ('stiff call, poses', s)
        if 'xfeat_err' in s:
            assert s['xfeat_err'] <= 2e-4 and s['xb2_err'] <= 3e-4, s

    # ---- one optimize_map call (Mapper.py:562-735)
    def on_map(self, rec, out):
        eng, cfg = self.eng, self.cfg
        rcfg = slam.render_cfg_from(cfg, cfg['rendering']['sigmoid_coef_mapper'])
        W = rec['W']
        knn, dpos = self._index(rec['pos'])
        N, R, iters = rec['pos'].shape[0], rec['R'], rec['iters']
        rows = rec['rows']
        n_geo = min(iters, rec['geo_iters'] + 1)                    # stage 'geometry' while it <= geo_iters (Mapper.py:594-597)

        def product(n_it=iters):
            dec = core.DecoderBlob(eng).pack(W)
            dgeo, dcol = eng.f32(rec['geo']).clone(), eng.f32(rec['col']).clone()
            xp = feats = mlp = None
            if rec['xfeats'] is not None:
                mlp = _exposure_module(W, eng.device)
                feats = [eng.f32(x).clone().requires_grad_(True) for x in rec['xfeats']]
                xp = (mlp, feats)
            mo = steps.MapOptimizer(eng, rcfg, dec, knn, dpos, dgeo, dcol, None, R, rec['lrs'], w_color=rec['w_color'], dynamic_radius=rec['rstack'] is not None,
                                    fix_color_decoder=rec['fix_color_decoder'], exposure=xp)
            assert mo._takes_native_loop()
            all_rows = rows.numel() == N and bool((rows == torch.arange(N)).all())
            if all_rows:
                mo.begin_frame()
            else:
                mask = torch.zeros(N, dtype=torch.uint8)
                mask[rows] = 1
                mo.new_frame(rows.to(torch.int32).to(eng.device), mask.to(eng.device))
            log = eng.zeros(n_it, 4)
            frames = (eng.f32(rec['dstack']), eng.f32(rec['cstack']), eng.f32(rec['pstack']), eng.f32(rec['rstack']) if rec['rstack'] is not None else None)
            mo.run(n_it, min(n_geo, n_it), frames, rec['rnd'][:n_it].to(torch.int32).to(eng.device), rec['fid'].to(torch.int32).to(eng.device), (0, self.H, 0, self.W),
                   self.intr, self.H, self.W, log)
            mo.finish()
            _sync(eng)
            return log, dgeo, dcol, dec, feats, mlp

        log, dgeo, dcol, dec, feats, mlp = product()
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
        # WHICH BRANCH: Adam's first step is sign-like (lr g / (|g| + 1e-8)) and the decoder's small tensors - the Fourier matrices - are shared by every
        # sample; their gradients are cancelling sums over the batch that fp32 resolves to ~1e-3 of the tensor's largest entry (the fp32 oracle
        # itself is that far from a float64 evaluation).  An entry below that takes its first step either way, and from the second iteration on
        # EVERY sample sees another embedding: on the chip about one 10-iteration TUM / ScanNet call in thirty ended with 1 % of its stepped
        # geometry entries 4e-3 ... 7e-3 from the oracle's (bulk of the other calls: 3e-4 ... 7e-4) - reproducibly on the same record (three
        # replays agree to 1e-4), on the host emulator as on the chip, per-statement path as native loop, exhaustive search as KD-tree, with ONE
        # entry of geo_decoder.embedder._B stepped the other way (oracle gradient +3.3e-4, product -1.2e-4, largest entry 0.65; float64: +1.3e-3;
        # tools/probe/tf_outlier.py).  So: the product's first step alone (one iteration of the same call) against the oracle's tensors after ITS
        # first step - the number of entries that went the other way is recorded and check_map holds a call to the tight bounds only on the
        # oracle's own branch.
        if iters <= 40 and iters > 1 and out.get('W_first'):
            W1 = product(1)[3].unpack()
            lr0 = rec['lrs']['geometry'][0]
            s['first_step_flips'] = int(sum(int(((W1[n].reshape(v.shape).cpu() - v).abs() > lr0).sum()) for n, v in out['W_first'].items() if n in W1)) if lr0 > 0 else 0
        short_outlier = iters <= 40 and (s['geo_err_q99'] > 0.02 * lr_g * iters ** 0.5 or s['geo_err_q999'] > 0.1 * lr_g * iters ** 0.5)
        if short_outlier:           # is it the call or the run?  the product twice more on the same record
            rp = []
            for _ in range(2):
                lg2, g2 = product()[:2]
                rp.append(dict(geo_err_q99=param_error_stats(g2.cpu()[rows], out['geo_rows'], rec['geo'][rows])['err_q99'],
                               geo_vs_first_run_q99=param_error_stats(g2.cpu()[rows], gk[rows], rec['geo'][rows])['err_q99']))
            s['replay'] = rp
            dump = os.environ.get('LK_TF_DUMP')                    # diagnosis: the call's record, for a replay elsewhere (tools/probe/tf_outlier.py)
            if dump:
                torch.save({'rec': {k: v for k, v in rec.items()}, 'out': out, 'case': self.case, 'product_geo_rows': gk[rows], 'product_losses': kl},
                           os.path.join(dump, f'tf_outlier_{self.case}_{rec["idx"]}.pt'))
        if iters > 40:
            # YARDSTICK of a long call: the oracle loop once more with both feature tables perturbed by 1e-7 relative, against the oracle's own run
            g = torch.Generator().manual_seed(1000 + rec['idx'])
            pert = lambda x: x * (1 + 1e-7 * torch.randn(x.shape, generator=g))
            render = L.TreeRender(rec['pos'], rcfg.rel_pos, near=rcfg.near_surface, far=rcfg.far_surface)
            render.cfg.radius_query, render.cfg.coef, render.cfg.min_nn = rcfg.radius_query, rcfg.coef, rcfg.min_nn
            if rec['xfeats'] is None:
                yl, yg, yc, _ = L.oracle_map_loop(render, pert(rec['geo']), pert(rec['col']), W, rows, (rec['dstack'], rec['cstack'], rec['pstack']), rec['fid'], rec['rnd'],
                                                  n_geo, self.intr, rec['lrs'], rec['dec_names'], w_color=rec['w_color'], rstack=rec['rstack'])
                s['yard_loss_rel'] = (np.abs(np.array(yl) - ol) / np.abs(ol)).tolist()
                s.update({f'yard_geo_{k}': v for k, v in param_error_stats(yg, out['geo_rows'], rec['geo'][rows]).items()})
                s.update({f'yard_col_{k}': v for k, v in param_error_stats(yc, out['col_rows'], rec['col'][rows]).items()})
        s['loss_rel'] = rel.tolist()
        if rel[:min(5, n_geo + 1)].max() > 5e-5 and rec['xfeats'] is None:
            # a call that leaves the oracle's losses within its first iterations (one in ~20 calls at 500 rays): which side?  The oracle loop once
            # more on the UNPERTURBED inputs with the contract's exhaustive search (knn_exact) instead of the KD-tree, first iterations only
            m = min(iters, 8)
            rx = L.TreeRender(rec['pos'], rcfg.rel_pos, near=rcfg.near_surface, far=rcfg.far_surface, exact=True)
            rx.cfg.radius_query, rx.cfg.coef, rx.cfg.min_nn = rcfg.radius_query, rcfg.coef, rcfg.min_nn
            xl = np.array(L.oracle_map_loop(rx, rec['geo'], rec['col'], W, rows, (rec['dstack'], rec['cstack'], rec['pstack']), rec['fid'], rec['rnd'][:m],
                                            n_geo, self.intr, rec['lrs'], rec['dec_names'], w_color=rec['w_color'], rstack=rec['rstack'])[0])
            s['anomaly_exact_search_vs_kernel'] = (np.abs(xl - kl[:m]) / np.abs(ol[:m])).tolist()
            s['anomaly_exact_search_vs_oracle'] = (np.abs(xl - ol[:m]) / np.abs(ol[:m])).tolist()
        self.map_stats.append(s)
        _record(self.case, map=self.map_stats)

    def check_map(self, s):
        """Measured on the chip (round 5): first ten iterations <= 7e-6 (Replica model) / 1.3e-4 (TUM model at 10 000 rays); 10-iteration calls end
        <= 1.3e-4 apart; the 285-372-iteration calls of config 1 end 1-10 % apart with 1 % of the stepped entries 0.2-0.3 apart - the
        perturbed-oracle yardstick of the same calls: 0.5-18 %, 0.19-0.27."""
        rel = np.array(s['loss_rel'])
        iters, n_geo = s['iters'], s['n_geo']
        lr_g, lr_c = self.cfg['mapping']['stage']['geometry']['geometry_lr'], self.cfg['mapping']['stage']['color']['color_lr']
        # tight only up to the FIRST 'color' iteration: from the second one on a call follows one of a few discrete loss branches (the colour
        # decoder's sign-like first Adam step on noise-level gradient entries, DESIGN 5 "Round 5": 1e-5 ... 1.2e-3 at bench size) - a TUM call
        # with four 'geometry' iterations was 3e-5 ... 1.3e-4 in its iterations 5-9 in seven runs and 5.02e-4 in the eighth; the whole-call
        # bounds below cover that part
        # One config-1 call in ~20 (500 rays, a refinement call whose loss is down at 1.2) left the oracle's losses in its THIRD iteration: 1.5e-6,
        # 6e-7, 1.6e-5, 4.1e-4, 4.8e-5, 2.9e-4 ... against a yardstick of <= 4e-6 - a level shift after the first Adam steps with spikes of one
        # sample's worth (1 / (5 R) = 4e-4) in the iterations whose draws hit the rows concerned, the geometry rows' version of the same
        # sign-like first step.  Iterations 0 and 1 stay tight (forward + loss, the first step); the window behind them takes one sample's worth
        # at small batches (such a call records the oracle loop with the exhaustive search beside it: on_map, 'anomaly_*')
        t5, t10 = min(5, n_geo + 1), min(10, n_geo + 1)
        assert np.isfinite(rel).all() and rel[:2].max() <= 2e-5 and rel[:t5].max() <= max(5e-5, 0.25 / s['rays']) and \
            rel[:t10].max() <= max(5e-4, 0.5 / s['rays']), ('first iterations', s)
        assert s['untouched_rows_equal'], s
        n_col = max(1, iters - n_geo)
        if 'yard_loss_rel' in s:
            yard, h = np.array(s['yard_loss_rel']), iters // 2
            assert rel.max() <= 5.0 * yard.max() + 1e-3 and np.median(rel[h:]) <= 5.0 * np.median(yard[h:]) + 1e-3, ('loss against the perturbed-oracle yardstick', s)
            for t in ('geo', 'col'):
                assert s[f'{t}_err_q99'] <= max(3.0 * s[f'yard_{t}_err_q99'], 1e-3) and s[f'{t}_err_q999'] <= max(3.0 * s[f'yard_{t}_err_q999'], 5e-3), (t, s)
        else:
            assert rel.max() <= 2e-3, ('loss, whole call', s)
            # on the oracle's own branch (no shared decoder entry took its first step the other way, on_map "WHICH BRANCH") the stepped rows are
            # held tightly; on another branch to a sanity band (measured there: q99 <= 7.1e-3, q99.9 <= 2.3e-2 after ten iterations), and such a
            # call must reproduce on its own record (no run-to-run effect hiding behind the branch)
            k = 1.0 if s.get('first_step_flips', 0) == 0 else 6.0
            assert s['geo_err_q99'] <= k * 0.02 * lr_g * iters ** 0.5 and s['geo_err_q999'] <= k * 0.1 * lr_g * iters ** 0.5, s
            assert s['col_err_q99'] <= k * (0.05 * lr_c * n_col ** 0.5 + 2e-3) and s['col_err_q999'] <= k * (0.2 * lr_c * n_col ** 0.5 + 4e-3), s
            assert s['decoder_excess_q999'] <= 2e-2, s
            for r in s.get('replay', []):
                assert r['geo_vs_first_run_q99'] <= 1e-3, ('a call off the oracle\'s branch must reproduce on its own record', s)
        assert s['geo_err_max'] <= 2.0 * lr_g * iters and s['col_err_max'] <= 2.0 * max(lr_c, lr_g) * iters, s
        if 'xfeat_err' in s:
            assert s['xfeat_err'] <= 3e-4 and s['xb2_err'] <= 3e-4 and s['keyframe_feats_constant'], s

    def check_all(self):
        for s in self.track_stats:
            # (at the configured rate only the first TWO iterations are held tightly: the product is not bit-reproducible from run to run - the
            # order of the gather's float atomics - and at 500 rays one run of four sees its trajectory leave the oracle's in the third
            # iteration already (4.8e-4, then 5e-3 in the fifth); iteration 0 is forward + loss, iteration 1 the first backward + Adam step +
            # forward; every later iteration is covered by the stiff call)
            self.check_track(s, exact_iters=2)
        for s in self.map_stats:
            self.check_map(s)


def _load(path, **over):
    cfg = copy.deepcopy(config.load_config(os.path.join(ROOT, path), os.path.join(ROOT, 'configs/point_slam.yaml')))
    for sec, kv in over.items():
        cfg[sec].update(kv)
    return cfg


# BASELINE config 1: 500 rays per iteration, the room config's iteration counts - with the first frame's 1 500 iterations cut to 300 (an
# oracle iteration at 500 rays is 0.03-0.08 s: 1 500 of them would be two minutes of every run for the same code path)
CFG1 = dict(tracking=dict(pixels=500), mapping=dict(pixels=500, iters_first=300, geo_iter_first=120, color_refine=False),
            data=dict(n_frames=10, motion='handheld', scene='furnished'))
# TUM / ScanNet at their own ray budgets, 10 iterations per call, three frames (two tracked, frames 0 and 2 mapped)
CFG_TUM = dict(tracking=dict(iters=10), mapping=dict(iters_first=10, geo_iter_first=3, iters=10, every_frame=2, keyframe_every=1, color_refine=False,
                                                       min_iter_ratio=1.0),
               data=dict(n_frames=4, motion='handheld', scene='furnished'))


def run_teacher_forced(eng, case, cfg, n_frames):
    reader = slam.SyntheticRoomDataset(cfg, 'cpu', n_frames)            # (the product's frame reader: the synthetic sequence, cropped as the config says)
    frames = [reader[i] for i in range(n_frames)]
    o = OS.OracleSLAM(cfg, frames)
    rp = Replayer(eng, cfg, case)
    o.on_track, o.on_map = rp.on_track, rp.on_map
    o.run(n_frames)
    rp.check_all()
    return o, rp


@pytest.mark.gpu
def test_config1_ten_frames_teacher_forced():
    cfg = _load('configs/Synthetic/room.yaml', **CFG1)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    o, rp = run_teacher_forced(make_engine('hip'), 'config1-500rays-10frames', cfg, 10)
    assert len(rp.track_stats) == 8 and len(rp.map_stats) == 3          # frames 2..9 tracked; frames 0, 5 and the last one (9) mapped
    assert all('stiff_loss_rel' in s for s in rp.track_stats) and any('yard_loss_rel' in s for s in rp.map_stats)
    # the oracle's own trajectory over these ten frames (with the first frame's mapping cut to 300 iterations it tracks about as well as the
    # one-step constant-speed prior, 0.56 cm: recorded, and held below dead reckoning)
    gt = torch.stack([f[3] for f in o.frames[:10]])
    ate, prior = OS.ate_rmse(o.est[:10], gt), OS.prior_only_metrics(gt)
    _record('config1-500rays-10frames', oracle_ate_cm=100 * ate, prior_only=prior)
    assert 100 * ate < max(prior['dead_reckoning_ate_cm'], 2 * prior['one_step_ate_cm'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,path', (('tum', 'configs/TUM_RGBD/freiburg1_desk.yaml'), ('scannet', 'configs/ScanNet/scene0000.yaml')))
def test_tum_scannet_budgets_teacher_forced(name, path):
    cfg = _load(path, **CFG_TUM)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    o, rp = run_teacher_forced(make_engine('hip'), f'{name}-fullrays-10it', cfg, 4)
    assert len(rp.track_stats) == 2 and len(rp.map_stats) >= 2
    assert rp.track_stats[0]['points'] > 10_000 and rp.map_stats[-1]['rays'] >= 9_000
    if name == 'scannet':
        assert all('xfeat_err' in s for s in rp.track_stats) and rp.map_stats[-1]['xfeat_moved'] > 1e-4


def test_teacher_forced_miniature_on_the_emulator():
    """The same replay on the host emulator at a miniature size (24 x 32 frames, 48 / 64 rays): the harness itself in the CPU suite."""
    cfg = _load('configs/Synthetic/room.yaml', tracking=dict(ignore_edge_W=2, ignore_edge_H=2, pixels=48, iters=6),
                mapping=dict(pixels=64, pixels_adding=400, iters=6, iters_first=12, geo_iter_first=4, every_frame=2, keyframe_every=2,
                             mapping_window_size=4, color_refine=False),
                pointcloud=dict(radius_add=0.12, radius_query=0.24, radius_min=0.06), data=dict(n_frames=4, motion='handheld', scene='furnished'))
    cfg['cam'].update(H=24, W=32, fx=26.0, fy=26.0, cx=15.5, cy=11.5)
    torch.set_num_threads(4)
    o, rp = run_teacher_forced(make_engine('emu'), 'emu-miniature', cfg, 4)
    assert len(rp.track_stats) == 2 and len(rp.map_stats) == 3          # frames 2, 3 tracked; frames 0, 2 and the last one (3) mapped
    _REPORT.clear()
