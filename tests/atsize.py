"""Scenarios and oracle evaluation for the parity tests at the BENCHMARKED sizes (tests/test_parity_at_size.py):
5 000 / 10 000-ray mapping batches and 1 500 / 5 000-ray tracking batches over a 100 000-point cloud (bench.py's
workload), forward spot checks at 2 M and 5 M points.  The oracle (oracle/hotpath.py, CPU autograd) finishes such a
batch in seconds once the neighbour search is done by its KD-tree form (hotpath.knn_tree) - rows where the fp64 tree
and the fp32 contract could order candidates differently are re-checked with the brute-force contract (knn_exact).

Test infrastructure only."""
import numpy as np
import torch

from oracle import hotpath as H
from loopy_slam_amd import synthetic as syn

I = syn.TUM_INTR
INTR = (I['fx'], I['fy'], I['cx'], I['cy'])
_SCENES = {}


def scene(N, seed=1219):
    """(pos, geo, col) CPU tensors of the synthetic room cloud, cached per size."""
    if (N, seed) not in _SCENES:
        _SCENES.clear()                     # one big cloud at a time (5 M points = 1.3 GB of features)
        _SCENES[(N, seed)] = syn.build_cloud(N, device='cpu', seed=seed)
    return _SCENES[(N, seed)]


def ray_batch(R, frame=7, holes=0.0, seed=0, window=None):
    """R random pixels of synthetic frame `frame`: dict(i, j, rays_o, rays_d, gt_depth, gt_color, c2w) on the CPU."""
    depth, color, c2w = syn.render_frame(frame, device='cpu', holes=holes)
    g = torch.Generator().manual_seed(seed * 7919 + R)
    H0, H1, W0, W1 = window if window is not None else (0, I['H'], 0, I['W'])
    i = torch.randint(W0, W1, (R,), generator=g).float()
    j = torch.randint(H0, H1, (R,), generator=g).float()
    ro, rd = syn.pixel_rays(c2w, i, j)
    gd = depth[j.long(), i.long()].contiguous()
    gc = color[j.long(), i.long()].contiguous()
    return dict(i=i, j=j, rays_o=ro, rays_d=rd, gt_depth=gd, gt_color=gc, c2w=c2w)


def ocfg(rel_pos):
    return H.RenderCfg(S=5, near_surface=0.98, far_surface=1.02, near_end=0.3, coef=0.1, k=8, min_nn=2,
                       radius_query=0.08, rel_pos=rel_pos)


def contract_knn(pos, p, r2, got_idx=None):
    """The kNN contract at size: KD-tree proposal re-ranked in fp32 (hotpath.knn_tree); where `got_idx` (the kernel's answer)
    disagrees, the rows are recomputed with the brute-force statement of the contract (hotpath.knn_exact).
    Returns (d2, idx, count, n_rechecked)."""
    d2, idx, cnt = H.knn_tree(pos.numpy(), p.numpy(), 8, r2)
    n_re = 0
    if got_idx is not None:
        bad = np.nonzero((np.asarray(got_idx) != idx).any(1))[0]
        n_re = int(bad.size)
        if n_re:
            assert n_re <= 2000, f'{n_re} rows differ from the KD-tree answer: not a tie-break effect'
            rr = r2 if np.ndim(r2) == 0 else np.asarray(r2)[bad]
            ed, ei, ec = H.knn_exact(pos.numpy(), p.numpy()[bad], 8, rr)
            d2[bad], idx[bad], cnt[bad] = ed, ei, ec
    return d2, idx, cnt, n_re


class ref64:
    """Context: the oracle's graph evaluated in FLOAT64 ON THE FP32-ROUNDED INPUTS - the referee between two fp32 implementations.
    Inside it, hotpath.sample_z / sample_points / fourier return the values an fp32 evaluation produces (sample depths, sample
    positions p = o + d z with the rounded multiply and add, Fourier arguments (2 pi x) @ B in torch-CPU's fp32 order - the quantities
    the kernels reproduce BIT FOR BIT) as float64 tensors whose derivative is the float64 graph's (value of the rounded quantity,
    derivative of the exact one: x32.double() + (x64 - x64.detach())); everything downstream - interpolation weights, both MLPs, the
    composite, the losses, autograd - runs in float64.  err(HIP, ref64) and err(fp32 oracle, ref64) are then both pure COMPUTATION
    errors of the same function of the same numbers, and the bound "HIP is no farther from float64 than the fp32 oracle is" does not
    depend on which SIMD kernels the host's torch picks."""

    def __enter__(self):
        self.saved = (H.sample_z, H.sample_points, H.fourier)
        o_z, o_p, o_f = self.saved

        def sample_z(gt_depth, *a):
            z, far = o_z(gt_depth.detach().float(), *a)
            return z.double(), far

        def sample_points(ro, rd, z):
            p32 = o_p(ro.detach().float(), rd.detach().float(), z.detach().float())
            pg = o_p(ro.double(), rd.double(), z.double())
            return p32.double() + (pg - pg.detach())

        def fourier(x, B, concat):
            y32 = H.fourier_arg(x.detach().float(), B.detach().float())          # (the kernels' fma sequence, whatever matmul kernel this host's MKL picks)
            yg = (H.TWO_PI * x.double()) @ B.double()
            y = y32.double() + (yg - yg.detach())
            return torch.cat((torch.sin(y), torch.cos(y)), dim=-1) if concat else torch.sin(y)
        H.sample_z, H.sample_points, H.fourier = sample_z, sample_points, fourier
        return self

    def __exit__(self, *exc):
        H.sample_z, H.sample_points, H.fourier = self.saved
        return False


def to64(x):
    """float tensors / dicts of them / kNN triples -> float64 (the VALUES stay the fp32 ones)."""
    if isinstance(x, dict):
        return {k: to64(v) for k, v in x.items()}
    if isinstance(x, tuple):
        return tuple(to64(v) for v in x)
    if torch.is_tensor(x):
        return x.double() if x.is_floating_point() else x
    if isinstance(x, np.ndarray):
        return x.astype(np.float64) if x.dtype.kind == 'f' else x
    return x


def oracle_mapper64(rel_pos, stage, b, pos, geo, col, W, knn, w_color=0.1, exclude=None):
    """oracle_mapper's gradients from the float64 referee (ref64)."""
    with ref64():
        return oracle_mapper(rel_pos, stage, to64(b), pos.double(), geo.double(), col.double(), to64(W), to64(tuple(knn)), w_color, True, exclude)


def oracle_tracker64(rel_pos, b, cam, pos, geo, col, W, knn, w_color=0.5, exclude=None, var32=None, rays_value=None):
    """var32: the fp32 evaluation's rendered variance.  The tracker's loss divides by sqrt(var) with var DETACHED (Tracker.py:171-175) - a
    constant of the differentiated function, and as a difference of nearly equal numbers the one quantity whose fp32 value is far (1e-3
    relative on flat rays) from its float64 value in EVERY fp32 implementation alike; the referee takes it as the rounded input it is."""
    with ref64():
        return oracle_tracker(rel_pos, to64(b), cam.double(), pos.double(), geo.double(), col.double(), to64(W), to64(tuple(knn)), w_color, exclude,
                              var_const=None if var32 is None else var32.double(), rays_value=rays_value)


def oracle_mapper(rel_pos, stage, b, pos, geo, col, W, knn, w_color=0.1, grads=True, exclude=None):
    """Oracle forward + mapper loss (+ autograd).  Returns dict(out=render dict, loss=(loss, geo, col, mask),
    g_geo, g_col, gW{name: grad}).  exclude: bool [R] rays left out of the loss that is differentiated (their loss
    gradient is zeroed on the kernel side too): rays on a branch point of the graph, see branch_point_rays."""
    names = [k for k in W if k != 'color_decoder.embedder._B']
    Wr = {k: v.clone().requires_grad_(grads and k in names) for k, v in W.items()}
    geo_r, col_r = geo.clone().requires_grad_(grads), col.clone().requires_grad_(grads)
    o = H.render_batch(ocfg(rel_pos), b['rays_o'], b['rays_d'], b['gt_depth'], pos, geo_r, col_r, Wr, stage, knn=knn)
    loss = H.mapper_loss(o['depth'], o['color'], o['valid_ray'], b['gt_depth'], b['gt_color'], stage, w_color)
    res = dict(out=o, loss=loss)
    if grads:
        lb = loss if exclude is None else H.mapper_loss(o['depth'], o['color'], o['valid_ray'] & ~exclude, b['gt_depth'], b['gt_color'], stage, w_color)
        lb[0].backward()
        res['g_geo'] = geo_r.grad
        res['g_col'] = col_r.grad if col_r.grad is not None else torch.zeros_like(col)
        res['gW'] = {k: Wr[k].grad for k in names if Wr[k].grad is not None}
    return res


def oracle_tracker(rel_pos, b, cam, pos, geo, col, W, knn, w_color=0.5, exclude=None, var_const=None, rays_value=None):
    """Oracle tracking iteration: rays from the 7-vector pose, render in tracker mode, tracker loss, autograd to the pose
    (and to the rays).  knn must be the list for these rays.
    rays_value = (rays_o, rays_d): the VALUES of the rays (the kernel's own fp32 rays) under the pose function's derivative - the
    gradient with respect to a sample position goes through 2 pi B cos(2 pi p B) with |B| ~ 25-32, sums of 93 / 40 terms that cancel: a
    one-ulp difference in a ray direction moves it by 1e-3 of its size, so two implementations are compared AT THE SAME RAYS."""
    cam_r = cam.clone().requires_grad_(True)
    ro, rd = H.rays_from_uv(b['i'], b['j'], H.quat_to_c2w(cam_r), *INTR)
    if rays_value is not None:
        ro = rays_value[0].to(ro.dtype) + (ro - ro.detach())
        rd = rays_value[1].to(rd.dtype) + (rd - rd.detach())
    ro, rd = ro.contiguous(), rd.contiguous()
    ro.retain_grad(); rd.retain_grad()
    o = H.render_batch(ocfg(rel_pos), ro, rd, b['gt_depth'], pos, geo, col, W, 'color', tracker=True, knn=knn)
    var = o['var'] if var_const is None else var_const
    loss = H.tracker_loss(o['depth'], var, o['color'], b['gt_depth'], b['gt_color'], w_color)
    if exclude is None:
        loss[0].backward()
    else:           # the same loss with the excluded rays' terms removed (mask and mean are those of the full batch)
        m = loss[3] & ~exclude
        tmp = torch.abs(b['gt_depth'] - o['depth']) / torch.sqrt(var.detach() + 1e-10)
        (torch.clamp(tmp, min=0.0, max=1e3)[m].sum() + w_color * torch.abs(b['gt_color'] - o['color'])[m].sum()).backward()
    return dict(out=o, loss=loss, g_cam=cam_r.grad, g_rays_o=ro.grad, g_rays_d=rd.grad, rays_o=ro.detach(), rays_d=rd.detach())


def errs(a, b):
    """(max-norm relative error, worst element-wise error relative to |b| + 1e-3 max|b|)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    return float(np.abs(a - b).max() / scale), float((np.abs(a - b) / (np.abs(b) + 1e-3 * scale)).max())


def geo_gate_margin(out, pos, geo, W, r2=None):
    """min over the geometry decoder's ReLU units of |pre-activation|, per sample (decoder.py:274-288).  A sample whose
    margin is at rounding level can take the other branch of a ReLU in ANY second fp32 implementation: its gradient
    contribution then differs by a whole term, not by rounding.  The feature rows such samples touch are excluded from
    the element-wise gradient comparison (and counted)."""
    import torch.nn.functional as F
    with torch.no_grad():
        r2 = torch.tensor(np.float32(0.08 ** 2)) if r2 is None else r2
        c, _ = H.interpolate(out['p'], pos, geo, out['idx'], out['d2'], r2, out['count'], 2, torch.zeros(geo.shape[1]), False)
        e = H.fourier(out['p'], W['geo_decoder.embedder._B'], concat=False)
        h, margin = e, torch.full((e.shape[0],), float('inf'))
        for i in range(5):
            pre = F.linear(h, W[f'geo_decoder.pts_linears.{i}.weight'], W[f'geo_decoder.pts_linears.{i}.bias'])
            margin = torch.minimum(margin, pre.abs().min(dim=1).values)
            h = F.relu(pre) + F.linear(c, W[f'geo_decoder.fc_c.{i}.weight'], W[f'geo_decoder.fc_c.{i}.bias'])
            if i == 2:
                h = torch.cat([e, h], -1)
    return margin


def rows_of_samples(out, sample_mask):
    """Cloud rows (neighbours) of the flagged samples."""
    idx = out['idx'][sample_mask]
    return torch.unique(idx[idx >= 0].long())


def branch_point_rays(out, b, pos, geo, W, tracker_loss=None, tol=1e-5, r2=None):
    """bool [R]: rays whose gradient is not comparable between two fp32 implementations because they sit on a branch point
    of the graph at rounding level - a geometry-decoder ReLU with |pre-activation| < tol in one of their samples
    (geo_gate_margin), an L1 term with |depth - gt| < tol, or (tracker) a residual within 1e-5 relative of the loss mask's
    threshold 10 * mean or of the 1e3 clamp (Tracker.py:177-183).  Their loss gradient is zeroed on both sides; the count is
    recorded and bounded by the tests.
    tol = 1e-5 (round 4; 2e-6 before): the kernels' pre-activations differ from the CPU oracle's by ~1e-6, and the oracle's own move at that
    level with the host's SIMD kernels - at 2e-6 ONE ray within 2-3e-6 of a gate was classified differently on one box of the pool and its
    flipped term (7e-5 of the largest entry) failed the float64 referee's 2e-5 floor; 1e-5 keeps an order of magnitude between the
    rounding noise and the line."""
    R = b['gt_depth'].shape[0]
    margin = geo_gate_margin(out, pos, geo, W, r2=r2)          # r2: per-sample squared radius [P] (dynamic radius) or None = 0.08^2
    rays = (margin < tol).reshape(R, -1).any(1)
    rays |= (out['depth'].detach() - b['gt_depth']).abs() < tol
    if tracker_loss is not None:
        # tracker mode recomputes D from the positions and drops neighbours with D > r^2 (decoder.py:191-198): a neighbour within
        # rounding of the radius is in or out depending on the summation order of the squared distance
        r2e = torch.tensor(np.float32(0.08 ** 2)) if r2 is None else torch.as_tensor(r2).reshape(-1, 1)
        near_edge = ((out['d2'] - r2e).abs() < 4e-6 * r2e) & (out['idx'] >= 0)
        rays |= near_edge.any(1).reshape(R, -1).any(1)
        tmp = torch.abs(b['gt_depth'] - out['depth'].detach()) / torch.sqrt(out['var'].detach() + 1e-10)
        thr = 10 * tmp.mean()
        rays |= ((tmp - thr).abs() < 1e-5 * thr) | ((tmp - 1e3).abs() < 1e-2)
    return rays, float(margin.min())
