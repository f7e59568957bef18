"""The two per-frame loops AS THE BENCHMARK RUNS THEM - one lk_track_frame call of 40 iterations x 1 500 rays, one lk_map_frame call of
24 'geometry' + 36 'color' iterations x 5 000 rays on a frustum row list, N = 100 000 points, Replica model - against the oracle loop
(oracle/hotpath.py render + torch autograd + torch.optim.Adam: Tracker.py:313-401, Mapper.py:562-735) on the SAME draws.

Until round 4 the native loops (look-ahead search, pre-gathered and partitioned batches, step riders, loss folded into the decoder
launches, the weight-gradient fork) met the oracle loop only at 64-96 rays; at the bench's sizes the oracle comparisons went through the
per-statement path one iteration at a time.  A 1 500-ray tracking call is 250 tiles on 256 compute units with soft barriers and per-tile
partial sums, a 5 000-ray mapping call 782 + 196 workgroups on 768 slots over three streams - where races and stale reads would live.

The same tests run on the host emulator at a reduced size in the CPU suite (tests/test_loops_small_cpu.py).
Measured values: gpurun_out/loops_at_size.json."""
import json
import os

import numpy as np
import pytest
import torch

import atsize as A
from oracle import hotpath as H
from loopy_slam_amd import core, steps, synthetic as syn
from util import make_engine

pytestmark = pytest.mark.gpu
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
_REPORT = {}
MAP_LRS = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}          # mapping.stage.* (configs/point_slam.yaml:64-66)


def _record(case, **kv):
    _REPORT.setdefault(case, {}).update(kv)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'loops_at_size.json'), 'w') as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True, default=float)


def _sync(eng):
    if eng.device.type == 'cuda':
        torch.cuda.synchronize()


class TreeRender:
    """The oracle's render over a fixed cloud with its KD-tree neighbour search (hotpath.knn_tree: tree proposal re-ranked in fp32)."""

    def __init__(self, pos, rel_pos, near=0.98, far=1.02, exposure=False, exact=False):
        from scipy.spatial import cKDTree
        self.pos = pos
        self.exact = exact                  # hotpath.knn_exact (the contract itself, O(P N)) instead of the tree: diagnosis of a disagreement
        self.tree = cKDTree(pos.numpy().astype(np.float64))
        self.cfg = H.RenderCfg(S=5, near_surface=near, far_surface=far, near_end=0.3, coef=0.1, k=8, min_nn=2, radius_query=0.08,
                               rel_pos=rel_pos, exposure=exposure)

    def __call__(self, ro, rd, gd, geo, col, W, stage, tracker=False, r2_ray=None, affine=None, color_sigmoid=True):
        c = self.cfg
        z, _ = H.sample_z(gd, c.near_surface, c.far_surface, c.near_end, c.S)
        p = H.sample_points(ro.detach(), rd.detach(), z).numpy()
        r2 = np.float32(c.radius_query ** 2) if r2_ray is None else r2_ray.float().reshape(-1, 1).repeat(1, c.S).reshape(-1).numpy()
        kn = H.knn_exact(self.pos.numpy(), p, 8, r2) if self.exact else H.knn_tree(self.pos.numpy(), p, 8, r2, tree=self.tree)
        return H.render_batch(c, ro, rd, gd, self.pos, geo, col, W, stage, tracker=tracker, r2_ray=r2_ray, affine=affine,
                              color_sigmoid=color_sigmoid, knn=kn)


def oracle_track_loop(render, geo, col, W, cam0, depth, color, flat_all, win, intr, lr, separate, w_color=0.5, r2_map=None):
    """Tracker.py:313-401 on the oracle: returns (losses, masked counts, candidate poses, index of the chosen one)."""
    iters = flat_all.shape[0]
    if separate:
        q, T = cam0[:4].clone().requires_grad_(True), cam0[4:].clone().requires_grad_(True)
        opt = torch.optim.Adam([{'params': [T], 'lr': lr}, {'params': [q], 'lr': 0.2 * lr}])
    else:
        leaf = cam0.clone().requires_grad_(True)
        opt = torch.optim.Adam([{'params': [leaf], 'lr': lr}])
    w = win[3] - win[2]
    losses, masked, cands = [], [], []
    for it in range(iters):
        cam = torch.cat([q, T]) if separate else leaf
        if separate:
            cands.append(cam.detach().clone())                  # the concatenation made BEFORE the step (Tracker.py:363-377)
        opt.zero_grad()
        fl = flat_all[it].long()
        i, j = (win[2] + fl % w).to(cam.dtype), (win[0] + torch.div(fl, w, rounding_mode='floor')).to(cam.dtype)
        ro, rd = H.rays_from_uv(i, j, H.quat_to_c2w(cam), *intr)
        gd, gc = depth[j.long(), i.long()], color[j.long(), i.long()]
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        r2 = r2_map[j.long(), i.long()][keep] if r2_map is not None else None
        out = render(ro[keep], rd[keep], gd[keep], geo, col, W, 'color', tracker=True, r2_ray=r2)
        loss, _, _, m = H.tracker_loss(out['depth'], out['var'], out['color'], gd[keep], gc[keep], w_color)
        loss.backward()
        opt.step()
        if not separate:
            cands.append(leaf.detach().clone())
        losses.append(float(loss.detach()))
        masked.append(int(m.sum()))
    return losses, masked, cands, int(np.argmin(losses))


def envelope_ok(e_a, e_yard, factor, floor):
    """e_a[it] <= max(factor x the largest yardstick error up to iteration it, floor) for every it (errors of a chaotic trajectory grow with
    the iteration; a late spike of the yardstick's is allowed the other side too).  Returns (ok, bound)."""
    bound = np.maximum(factor * np.maximum.accumulate(np.asarray(e_yard)), floor)
    return bool((np.asarray(e_a) <= bound).all()), bound


def stiff_call_ok(rel, dm, rays, tight):
    """Per-iteration loss differences of a STIFF tracking call (rate / 200: rounding is not amplified).  Iterations in which both sides mask the
    same number of rays: `tight` (5e-5 at >= 1 500 rays; one sample on the radius edge is ~1 / (5 R)).  A ray within rounding of the loss
    mask's threshold, 10 x the batch mean of the residual (Tracker.py:177-183), is in on one side and out on the other - the single-iteration
    tests exclude such rays, a loop cannot - and carries ~10 / R of the geometry loss: measured 1.3e-2 at 500 rays, once in 24 stiff calls of
    40 iterations; allowed 25 / R per ray, in at most a tenth of the iterations, two rays at most."""
    rel, dm = np.asarray(rel), np.asarray(dm)
    same = dm == 0
    return bool((rel[same] <= tight).all() and (rel[~same] <= 25.0 * dm[~same] / rays).all() and (~same).sum() <= max(2, rel.size // 10) and dm.max() <= 2)


def run_track_case(eng, case, N, R, iters, rel_pos, separate, lr, grad_pool=False, dynamic=False, exact_iters=3):
    """One lk_track_frame call against oracle_track_loop.  grad_pool: the pixels come from the pool of the highest colour-gradient pixels
    (tracking.sample_with_color_grad, common.py:198-234), dynamic: per-pixel query radius (use_dynamic_radius) - both from the ORACLE's image
    pre-pass, handed to the product as inputs (the call under test is the loop, not the pre-pass).

    Two calls on the same draws.  Pose optimisation is CHAOTIC at the configured rate: the gradient goes through 2 pi B cos(2 pi p B) with
    |B| ~ 25-32, and two fp32 evaluations of the same loop - the product's native and per-statement paths, or the fp32 oracle and its
    float64 evaluation - separate by a factor of ~10 per iteration once they differ at all (measured: 1e-7 -> 1e-4 relative loss
    difference within six iterations at 160 rays), so at lr = cam_lr only the first iterations can be compared tightly.
      'stiff'  lr = cam_lr / 200: the pose moves by < 1e-5 per iteration, rounding differences are not amplified, and EVERY iteration of the
               call is compared on (all but) identical inputs: loss 5e-5, masked-ray counts exactly, every candidate pose to 2 % of one
               step, the chosen pose - the whole launch sequence over the whole call (pose ring, look-ahead pose step, per-tile partial
               sums, Adam moments through the step direction).
      'config' lr = cam_lr: the first `exact_iters` iterations as tightly, then the trajectory stays in the oracle's neighbourhood
               (loss within 5 %, pose within a few steps) and optimises as well as the oracle's does."""
    pos, geo, col = A.scene(N)
    W = syn.default_weights(rel_pos=rel_pos)
    depth, color, c2w = syn.render_frame(5, device='cpu', holes=0.02)
    Hh, Ww = depth.shape
    intr = A.INTR
    cam0 = H.c2w_to_cam(c2w) + torch.tensor([0.0, 0.002, -0.001, 0.0015, 0.004, -0.003, 0.002])
    g = torch.Generator().manual_seed(1500 + R)
    r2_map = None
    e = 100 if not grad_pool else 20
    win = (e, Hh - e, e, Ww - e)
    if grad_pool or dynamic:
        grad, _, r_query = H.radius_maps(color.numpy(), 0.08, 0.02, 2, 0.15)
        if dynamic:
            r2_map = (torch.from_numpy(r_query).double() ** 2).float()
    if grad_pool:
        pool = torch.from_numpy(H.top_grad_pixels(grad, 15 * R, win, depth.numpy(), False)).long()
        u = torch.rand(iters, pool.numel(), generator=g)
        flat_all = pool[u.topk(min(R, int(pool.numel())), dim=1).indices]
        win = (0, Hh, 0, Ww)
    else:
        flat_all = torch.randint(0, (win[1] - win[0]) * (win[3] - win[2]), (iters, R), generator=g, dtype=torch.int32).long()
    render = TreeRender(pos, rel_pos)
    cfg = core.RenderCfg(rel_pos=rel_pos)
    dec = core.DecoderBlob(eng).pack(W)
    dpos, dgeo, dcol = eng.f32(pos), eng.f32(geo), eng.f32(col)
    knn = core.KnnIndex(eng, capacity=N)
    knn.build(dpos)
    dflat, ddepth, dcolor = flat_all.to(torch.int32).to(eng.device), eng.f32(depth), eng.f32(color)
    dr2 = eng.f32(r2_map) if r2_map is not None else None
    results = []
    for mode, lr_m in (('stiff', lr / 200.0), ('config', lr)):
        o_losses, o_masked, o_cands, o_best = oracle_track_loop(render, geo, col, W, cam0, depth, color, flat_all, win, intr, lr_m, separate, r2_map=r2_map)
        to = steps.TrackOptimizer(eng, cfg, dec, knn, dpos, dgeo, dcol, flat_all.shape[1], lr_m, separate_lr=separate, w_color=0.5, dynamic_radius=dynamic)
        assert to.native_loop
        best, log = to.track(eng.f32(cam0), ddepth, dcolor, iters, win, intr, dflat, r2_map=dr2)
        _sync(eng)
        k_losses, k_masked = log[:, 0].cpu().numpy().astype(np.float64), log[:, 3].cpu().numpy().astype(int)
        hist = to._keep_native[5].cpu()                      # the candidate pose of every iteration (lk_track_desc::hist)
        ol = np.array(o_losses)
        rel = np.abs(k_losses - ol) / np.abs(ol)
        dm = np.abs(k_masked - np.array(o_masked))
        # poses are compared as [R | t] matrices: the quaternion is NOT normalised (common.py:301-343 divides by |q|^2), so the gradient along
        # q's own direction is exactly zero analytically and rounding noise numerically - Adam's sign-like step on that component moves |q|
        # by up to lr per iteration either way, in every implementation, without moving the pose
        c2w = lambda c: H.quat_to_c2w(c.double())
        perr = np.array([float((c2w(hist[it]) - c2w(o_cands[it])).abs().max()) for it in range(iters)])
        k_best = int(np.argmin(k_losses))
        pose_err = float((c2w(best.cpu()) - c2w(o_cands[o_best])).abs().max())
        _record(f'{case}-{mode}', lr=lr_m, loss_rel=rel.tolist(), masked_diff=dm.tolist(), pose_err=perr.tolist(), chosen_iteration=(k_best, o_best),
                chosen_pose_err=pose_err, loss_first=float(ol[0]), loss_best=float(ol.min()), loss_best_hip=float(k_losses.min()), rays=int(flat_all.shape[1]),
                iters=iters, moved=float((o_cands[-1] - cam0).abs().max()))
        results.append((mode, lr_m, rel, dm, perr, pose_err, k_best, o_best, k_losses, ol, o_cands))
    for mode, lr_m, rel, dm, perr, pose_err, k_best, o_best, k_losses, ol, o_cands in results:
        assert np.isfinite(k_losses).all()
        n = iters if mode == 'stiff' else min(exact_iters, iters)
        # measured on the chip (1 500 x 40, stiff): loss <= 1.3e-6, masked counts equal, poses <= 6e-8 over all 40 iterations
        if mode == 'stiff':
            assert stiff_call_ok(rel, dm, R, 5e-5), (case, mode, rel.tolist(), dm.tolist())
        else:
            assert stiff_call_ok(rel[:n], dm[:n], R, 5e-5), (case, mode, rel.tolist(), dm.tolist())
        # candidate poses: within 5 % of the distance one Adam step covers (sign-like first steps), per iteration
        assert (perr[:n] <= 0.05 * lr_m * (1 + np.arange(n)) + 1e-7).all(), (case, mode, perr.tolist())
        if mode == 'stiff':
            assert float((o_cands[-1] - cam0).abs().max()) > 0.5 * lr_m * (iters - 1) * (0.2 if separate else 1.0) * 0.5        # the steps are there
            assert pose_err <= 0.05 * lr_m * iters + 1e-7 and (k_best == o_best or abs(k_losses[k_best] - k_losses[o_best]) <= 1e-4 * abs(ol[o_best])), (case, pose_err, k_best, o_best)
        else:
            # (measured at 1 500 x 40: loss differences up to 6.5e-3, poses up to 0.017 = 8.5 steps apart after 40 iterations; at 5 000 x 10 with
            # one leaf tensor: 0.014 = 7 steps after 10; both trajectories reach the same lowest loss to 1e-3)
            assert rel.max() <= 5e-2 and dm.max() <= max(3, R // 200), (case, mode, rel.tolist(), dm.tolist())
            assert perr.max() <= lr_m * iters, (case, mode, perr.tolist())
            assert abs(k_losses.min() / ol.min() - 1) <= 2e-2                  # the lowest loss of the call (what selects the pose): as the oracle's


def oracle_map_loop(render, geo, col, W, rows, frames, fid, rnd_all, n_geo, intr, lrs, dec_names, w_color=0.1, rstack=None):
    """Mapper.py:562-735 on the oracle (fresh Adam over {decoders, geometry rows, colour rows}): returns (losses, geo_p, col_p, W after)."""
    depth_s, color_s, pose_s = frames
    F = depth_s.shape[0]
    Ww = depth_s.shape[2]
    fx, fy, cx, cy = intr
    Wt = {k: v.clone() for k, v in W.items()}
    for n in dec_names:
        Wt[n].requires_grad_(True)
    geo_p, col_p = geo[rows].clone().requires_grad_(True), col[rows].clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [Wt[n] for n in dec_names], 'lr': 0}, {'params': [geo_p], 'lr': 0}, {'params': [col_p], 'lr': 0}])
    dflat, cflat = depth_s.reshape(F, -1), color_s.reshape(F, -1, 3)
    losses = []
    for it in range(rnd_all.shape[0]):
        stage = 'geometry' if it < n_geo else 'color'
        for gi in range(3):
            opt.param_groups[gi]['lr'] = lrs[stage][gi]
        opt.zero_grad()
        geo_t, col_t = geo.index_put((rows,), geo_p), col.index_put((rows,), col_p)
        fl = rnd_all[it].long()
        i, j = (fl % Ww).float(), torch.div(fl, Ww, rounding_mode='floor').float()
        dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
        Rm = pose_s[fid]
        rd = torch.sum(dirs[:, None, :] * Rm[:, :3, :3], -1)
        ro = Rm[:, :3, 3]
        gd, gc = dflat[fid, fl], cflat[fid, fl]
        keep = gd > 0
        keep = keep & (gd <= H.inside_threshold(gd[keep]))
        r2 = rstack.reshape(F, -1)[fid, fl][keep] if rstack is not None else None
        out = render(ro[keep], rd[keep], gd[keep], geo_t, col_t, Wt, stage, r2_ray=r2)
        loss = H.mapper_loss(out['depth'], out['color'], out['valid_ray'], gd[keep], gc[keep], stage, w_color)[0]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, geo_p.detach(), col_p.detach(), {n: Wt[n].detach() for n in dec_names}


def param_error_stats(mine, ref, before):
    """How far two fp32 Adam trajectories of a tensor are apart, relative to what the tensor moved: the bulk (99 % of the entries), the tail
    (99.9 %) and the largest entry - Adam's first steps are sign-like (lr g / (|g| + 1e-8)), so an entry whose gradient is rounding noise in an
    iteration steps the other way by up to 2 lr."""
    err = (mine.double() - ref.double()).abs().reshape(-1)
    moved = (ref.double() - before.double()).abs().reshape(-1)
    k = min(err.numel(), 4_000_000)
    q = lambda x, p: float(torch.quantile(x[:k].float(), p)) if x.numel() > 1000 else float(x.max())
    return dict(err_q99=q(err, 0.99), err_q999=q(err, 0.999), err_max=float(err.max()), moved_q50=q(moved, 0.5), moved_max=float(moved.max()))


def run_map_case(eng, case, N, R, iters, n_geo, rel_pos, window=12, exact_iters=5):
    """One lk_map_frame call (MapOptimizer.run, native loop) against oracle_map_loop on the frustum rows of the mapped frame.
    The mapper's trajectory is NOT chaotic in its losses (no pose in the loop: measured 1e-7 ... 3e-6 relative over the call at 240 rays), so
    the per-iteration losses are compared tightly over the whole call.  The parameters are compared three ways: rows outside the list bit for
    bit; entry by entry in quantiles (Adam's steps are sign-like, lr g / (|g| + 1e-8): an entry whose gradient is of the order of the
    rounding noise steps either way, and the kernels' split products carry an ABSOLUTE error floor where fp32 carries a relative one, so
    the tail of low-sensitivity entries is wider than between two fp32 evaluations - measured 4e-4 against 2e-5 at the 99 % level after
    ten iterations); and FUNCTIONALLY: a held-out batch rendered by the oracle from the product's optimised map against the same render from
    the oracle's optimised map.  Everything past the first iterations is held against a YARDSTICK: the oracle loop re-run with its feature
    tables perturbed by 1e-7 relative - the product may be 3 x as far from the oracle as that run is."""
    pos, geo, col = A.scene(N)
    W = syn.default_weights(rel_pos=rel_pos)
    fr = [syn.render_frame(3 * k, device='cpu', holes=0.02) for k in range(window)]
    depth_s, color_s, pose_s = (torch.stack([f[q] for f in fr]).contiguous() for q in range(3))
    Hh, Ww = depth_s.shape[1:]
    intr = A.INTR
    g = torch.Generator().manual_seed(5000 + R)
    rnd_all = torch.randint(0, Hh * Ww, (iters, R), generator=g, dtype=torch.int32)
    fid = (torch.arange(R) % window).long()
    rows = torch.from_numpy(H.frustum_rows(pos.numpy(), pose_s[0].numpy(), depth_s[0].numpy(), *intr, Hh, Ww, -4)).long()
    assert 0.02 * N < rows.numel() < 0.6 * N
    dec_names = list(steps.GEO_DECODER_PARAMS) + [n for n in steps.COLOR_DECODER_PARAMS if n in W and (rel_pos or ('mlp_col_neighbor' not in n and 'embedder_rel_pos' not in n))]
    render = TreeRender(pos, rel_pos)
    o_losses, geo_o, col_o, W_o = oracle_map_loop(render, geo, col, W, rows, (depth_s, color_s, pose_s), fid, rnd_all, n_geo, intr, MAP_LRS, dec_names)
    # YARDSTICK: the oracle loop once more with both feature tables perturbed by 1e-7 relative - how far two fp32 evaluations of THIS call
    # drift apart by themselves (the 'color' stage trains the decoder every sample shares: rounding differences feed back)
    gp = torch.Generator().manual_seed(3)
    pert = lambda x: x * (1 + 1e-7 * torch.randn(x.shape, generator=gp))
    y_losses, geo_y, col_y, W_y = oracle_map_loop(render, pert(geo), pert(col), W, rows, (depth_s, color_s, pose_s), fid, rnd_all, n_geo, intr, MAP_LRS, dec_names)
    cfg = core.RenderCfg(rel_pos=rel_pos)
    dec = core.DecoderBlob(eng).pack(W)
    dpos, dgeo, dcol = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
    knn = core.KnnIndex(eng, capacity=N)
    knn.build(dpos)
    drows = rows.to(torch.int32).to(eng.device)
    mask = torch.zeros(N, dtype=torch.uint8)
    mask[rows] = 1
    mo = steps.MapOptimizer(eng, cfg, dec, knn, dpos, dgeo, dcol, None, R, MAP_LRS, w_color=0.1)
    assert mo._takes_native_loop()
    mo.new_frame(drows, mask.to(eng.device))
    log = eng.zeros(iters, 4)
    frames = (eng.f32(depth_s), eng.f32(color_s), eng.f32(pose_s), None)
    mo.run(iters, n_geo, frames, rnd_all.to(eng.device), fid.to(torch.int32).to(eng.device), (0, Hh, 0, Ww), intr, Hh, Ww, log)
    _sync(eng)
    k_losses = log[:, 0].cpu().numpy().astype(np.float64)
    rel = np.abs(k_losses - np.array(o_losses)) / np.abs(np.array(o_losses))
    yrel = np.abs(np.array(y_losses) - np.array(o_losses)) / np.abs(np.array(o_losses))
    _record(case, loss_rel=rel.tolist(), yard_loss_rel=yrel.tolist(), loss_first=o_losses[0], loss_last=o_losses[-1], rows=int(rows.numel()), rays=R, iters=iters, n_geo=n_geo)
    checks = []          # (every statistic is recorded before the first assertion: a failing run leaves its numbers in the report)
    checks.append((bool(np.isfinite(k_losses).all()), 'finite losses'))
    # measured on the chip (5 000 x 60): <= 2e-6 through the 24 'geometry' iterations, then growing by ~15 % per iteration of the 'color' stage
    # to 1-4e-4 (Adam's sign-like steps on noise-level entries feed back through the colour decoder's weights)
    checks.append((rel[:exact_iters].max() <= 2e-5 and rel[:n_geo].max() <= 5e-5 and rel[min(n_geo, iters - 1)] <= 5e-5, ('loss, first iterations / geometry stage', rel.tolist())))
    # (the call has BRANCHES: 24 runs of it on one box - tools/probe/split_step_race.py, profiles/r5_map_call_branches.txt - follow one of four
    # loss sequences from the second 'color' iteration on, e.g. 9.3e-6, 5.9e-6, 9.7e-6 ... or 2.7e-4, 2.3e-4, 2.2e-4 ..., the same digits every
    # time: which one is decided by the sign of a few noise-level gradient entries in the colour decoder's first, sign-like Adam step, i.e. by
    # the order of the gather's float atomics in the 24 'geometry' iterations before it; the largest difference of a call was 3.8e-4 ... 1.2e-3)
    checks.append((rel.max() <= max(3.0 * yrel.max(), 3e-3), ('loss, whole call, against the perturbed-oracle yardstick', rel.tolist(), yrel.tolist())))
    if iters - n_geo >= 20:           # (every iteration draws its own batch: a trend needs a few of them)
        checks.append((o_losses[-1] < o_losses[min(n_geo, iters - 1)], 'the colour stage lowered its loss'))
    # rows outside the list: bit for bit where they were
    other = torch.ones(N, dtype=torch.bool)
    other[rows] = False
    gk, ck = dgeo.cpu(), dcol.cpu()
    checks.append((torch.equal(gk[other], geo[other]) and torch.equal(ck[other], col[other]), 'rows outside the list untouched'))
    n_it = {'geo': iters, 'col': iters - n_geo}
    for name, mine, ref, yard, before, lr in (('geo', gk[rows], geo_o, geo_y, geo[rows], 0.03), ('col', ck[rows], col_o, col_y, col[rows], 0.005)):
        st, sy = param_error_stats(mine, ref, before), param_error_stats(yard, ref, before)
        _record(case, **{f'{name}_rows_{k}': v for k, v in st.items()}, **{f'{name}_rows_yard_{k}': v for k, v in sy.items()})
        # 99 % / 99.9 % of the entries no farther from the oracle's than 3 x the perturbed oracle's are (floors: 2 % / 10 % of one lr step x
        # sqrt(iterations), a random walk of the noise-level entries); every entry within Adam's hard limit of 2 lr per iteration
        n = max(1, n_it[name]) ** 0.5
        checks.append((st['moved_max'] > 1e-3 and st['err_q99'] <= max(3.0 * sy['err_q99'], 0.02 * lr * n) and
                       st['err_q999'] <= max(3.0 * sy['err_q999'], 0.1 * lr * n) and st['err_max'] <= 2.0 * lr * n_it[name], (name + ' rows', st, sy)))
    Wk = {n: v.reshape(W_o[n].shape) for n, v in dec.unpack().items() if n in W_o}
    worst = {}
    for n in dec_names:
        if n not in Wk:
            continue
        st, sy = param_error_stats(Wk[n], W_o[n], W[n]), param_error_stats(W_y[n], W_o[n], W[n])
        scale = max(1.0, float(W[n].abs().max()))
        worst[n] = (st['err_q999'] / scale, st['err_max'] / scale, sy['err_q999'] / scale, sy['err_max'] / scale)
        # (floors relative to what the tensor MOVED.  A bias vector is 128 entries - its 99.9 % level IS its single worst entry, so below 1 024
        # entries the level gets the worst entry's floor: round 6 saw pts_linears.2.bias at 0.0086 with 0.057 moved on a host whose yardstick
        # run came out at 0.0013 - the same statistic of the yardstick is 0.004 ... 0.011 across the hosts of rounds 5-6)
        # (round 6, second host: pts_linears.0.weight - 5 120 entries - at 0.00995 with 0.0969 moved, 0.103 of it, on an unchanged mapper path: the
        # call's branches again; the level's floor for the large tensors is 0.15 of what the tensor moved)
        q_floor = 0.15 if Wk[n].numel() >= 1024 else 0.5
        checks.append((st['err_q999'] <= max(3.0 * sy['err_q999'], 2e-4 * scale + q_floor * st['moved_max']) and
                       st['err_max'] <= max(3.0 * sy['err_max'], 2e-4 * scale + 0.5 * st['moved_max']), (n, st, sy)))
    _record(case, decoder_err_q999_rel_max=max(v[0] for v in worst.values()), decoder_err_max_rel_max=max(v[1] for v in worst.values()),
            decoder_yard_q999_rel_max=max(v[2] for v in worst.values()), decoder_yard_max_rel_max=max(v[3] for v in worst.values()), decoder_tensors=len(worst))
    checks.append((len(worst) >= (28 if rel_pos else 23), 'every decoder tensor compared'))
    # functional comparison of the two optimised maps: a held-out batch of the mapped frame, oracle render from either parameter set
    ge = torch.Generator().manual_seed(77)
    fl = torch.randint(0, Hh * Ww, (min(4 * R, 4000),), generator=ge)
    i, j = (fl % Ww).float(), torch.div(fl, Ww, rounding_mode='floor').float()
    ro, rd = H.rays_from_uv(i, j, pose_s[0], *intr)
    gd = depth_s[0].reshape(-1)[fl]
    keep = gd > 0
    outs = []
    for g_rows, c_rows, Wd in ((gk[rows], ck[rows], Wk), (geo_o, col_o, W_o), (geo_y, col_y, W_y)):
        Wf = dict(W)
        Wf.update({n: Wd[n] for n in dec_names if n in Wd})
        with torch.no_grad():
            outs.append(render(ro[keep], rd[keep], gd[keep], geo.index_put((rows,), g_rows), col.index_put((rows,), c_rows), Wf, 'color'))
    e_d, e_c = A.errs(outs[0]['depth'], outs[1]['depth'])[0], A.errs(outs[0]['color'], outs[1]['color'])[0]
    y_d, y_c = A.errs(outs[2]['depth'], outs[1]['depth'])[0], A.errs(outs[2]['color'], outs[1]['color'])[0]
    dc = (outs[0]['color'] - outs[1]['color']).abs().max(1).values
    yc = (outs[2]['color'] - outs[1]['color']).abs().max(1).values
    q = lambda x, p: float(torch.quantile(x, p))
    moved_d = A.errs(outs[1]['depth'], gd[keep])[0]
    _record(case, functional_depth_rel=e_d, functional_color_rel=e_c, functional_yard_depth_rel=y_d, functional_yard_color_rel=y_c, functional_rays=int(keep.sum()),
            functional_color_q50=q(dc, .5), functional_color_q99=q(dc, .99), functional_yard_color_q50=q(yc, .5), functional_yard_color_q99=q(yc, .99),
            functional_depth_err_of_the_map=moved_d)
    # rendered depth / colour from the product's optimised map against the oracle's: no farther than 3 x the perturbed oracle's own render is
    # (worst ray and the 99 % level of the per-ray colour difference), floors 2e-4 / 1e-3
    checks.append((e_d <= max(3.0 * y_d, 2e-4) and e_c <= max(3.0 * y_c, 1e-3) and q(dc, .99) <= max(3.0 * q(yc, .99), 5e-4), ('functional', e_d, e_c, y_d, y_c)))
    failed = [c[1] for c in checks if not c[0]]
    _record(case, failed_checks=[str(f)[:200] for f in failed])
    assert not failed, (case, failed)
    return rel


def test_track_call_at_bench_size():
    """(a) 40 iterations x 1 500 rays, N = 100 000, Replica model, separate_LR: per-iteration losses, masked-ray counts and the chosen pose."""
    run_track_case(make_engine('hip'), 'track-replica-1500x40', 100_000, 1500, 40, True, True, 0.002)


def test_map_call_at_bench_size():
    """(b) 24 'geometry' + 36 'color' iterations x 5 000 rays over twelve keyframes on the frustum row list: losses, every updated feature
    row, every decoder tensor; untouched rows bit-equal."""
    run_map_case(make_engine('hip'), 'map-replica-5000x60', 100_000, 5000, 60, 24, True)


def test_map_call_tum_budget_without_relpos():
    """(b') the TUM mapping budget's ray count with the plain colour model (no rel-pos MLP): 6 'geometry' + 14 'color' iterations x 10 000
    rays.  Without the rel-pos launch the next iteration's interpolation rewrites the interpolated colour features right at its start -
    the buffer the split step's k_wgrad (still running on the side stream) streams as its fc_c columns (round-5 advisor; the fix and the
    deterministic form of the check: tests/test_split_step_order.py)."""
    run_map_case(make_engine('hip'), 'map-tum-10000x20', 100_000, 10000, 20, 6, False, window=10)


def test_track_call_tum_model_gradient_pool():
    """(c) the TUM / ScanNet tracker: 5 000 rays per iteration from the gradient-pixel pool, one leaf pose tensor (candidate AFTER the step),
    per-pixel dynamic query radius, plain colour model - 10 iterations."""
    # (exact_iters 3: one leaf tensor stepped at the full rate - the quaternion at lr instead of 0.2 lr - separates faster at the configured
    # rate: measured 1.4e-7, 5.8e-7, then 1.3e-4 in the fourth iteration; the stiff call covers all ten)
    run_track_case(make_engine('hip'), 'track-tum-5000x10', 100_000, 5000, 10, False, False, 0.002, grad_pool=True, dynamic=True, exact_iters=3)
