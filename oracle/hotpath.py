"""CPU restatement of the per-frame neural-point render/optimise hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch-CPU fp32, written from
the numerical specification in SURVEY.md Appendix A; every function cites the
reference lines (relative to /root/reference) whose behaviour it restates and is
pinned against golden vectors captured from the imported reference
(tests/golden/*.npz, tests/test_oracle_golden.py).

Conventions
-----------
R rays, S samples per ray, P = R*S query points (ray-major: point r*S+s),
k = 8 neighbours, C = 32 feature channels, N cloud points.
Weights are a flat dict keyed by the reference's state_dict names
('geo_decoder.pts_linears.0.weight', ...) plus 'color_decoder.embedder._B'
(a fixed random [3,20] matrix that the reference keeps outside its state_dict,
src/conv_onet/models/decoder.py:32).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi
FLT_MAX = float(np.finfo(np.float32).max)


# --------------------------------------------------------------------------- pose / rays
def quat_to_c2w(cam):
    """cam = (qw,qx,qy,qz, tx,ty,tz), quaternion NOT normalised -> [3,4].

    Restates get_camera_from_tensor / quad2rotation (src/common.py:301-343).
    """
    qr, qi, qj, qk = cam[0], cam[1], cam[2], cam[3]
    two_s = 2.0 / (cam[:4] * cam[:4]).sum()
    rows = [
        torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)]),
        torch.stack([two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr)]),
        torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)]),
    ]
    rot = torch.stack(rows)
    return torch.cat([rot, cam[4:7].reshape(3, 1)], dim=1)


def c2w_to_cam(c2w):
    """4x4 / 3x4 -> (qw,qx,qy,qz,T) float32.  get_tensor_from_camera (src/common.py:354-379)."""
    from scipy.spatial.transform import Rotation
    m = np.asarray(c2w, dtype=np.float64)
    q = Rotation.from_matrix(m[:3, :3]).as_quat()          # x,y,z,w
    q = np.roll(q, 1)                                       # w,x,y,z
    return torch.from_numpy(np.concatenate([q, m[:3, 3]])).float()


def rays_from_uv(i, j, c2w, fx, fy, cx, cy):
    """Pixel (i=column, j=row) -> (rays_o, rays_d), un-normalised directions.

    get_rays_from_uv (src/common.py:104-120).
    """
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs.reshape(-1, 1, 3) * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def image_rays(H, W, fx, fy, cx, cy, c2w, crop_edge=0):
    """All rays of an image, row-major.  get_rays (src/common.py:425-442)."""
    jj, ii = torch.meshgrid(
        torch.linspace(crop_edge, H - 1 - crop_edge, H - 2 * crop_edge),
        torch.linspace(crop_edge, W - 1 - crop_edge, W - 2 * crop_edge), indexing='ij')
    ro, rd = rays_from_uv(ii.reshape(-1), jj.reshape(-1), c2w, fx, fy, cx, cy)
    return ro, rd


def inside_threshold(depth):
    """min(10*median, 1.2*max) over the (positive) depths of a batch.

    Tracker.py:153-155, Mapper.py:674-676.  torch.median = lower median.
    """
    return torch.minimum(10 * depth.median(), 1.2 * depth.max())


# --------------------------------------------------------------------------- depth-guided samples
def sample_z(gt_depth, near_surface, far_surface, near_end, S):
    """Per-ray sample depths z[R,S] (Renderer.py:98-165, sample_near_pcl=False)."""
    gt = gt_depth.reshape(-1).float()
    far_bb = torch.minimum(5 * gt.mean(), torch.max(gt * 1.2))
    if torch.max(gt) > 0:
        far = torch.clamp(far_bb, 0, torch.max(gt * 1.2))
    else:
        far = far_bb
    t = torch.linspace(0.0, 1.0, steps=S)
    g = gt.reshape(-1, 1).repeat(1, S)
    z_nonzero = near_surface * g * (1. - t) + far_surface * g * t
    z_zero = torch.linspace(near_end, float(far), steps=S).reshape(1, S).expand(gt.shape[0], S)
    z = torch.where((gt > 0).reshape(-1, 1), z_nonzero, z_zero)
    return z, far


def sample_points(rays_o, rays_d, z):
    """p[r,s] = o + d*z (rounded multiply then rounded add; Renderer.py:167-169)."""
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    return pts.reshape(-1, 3)


# --------------------------------------------------------------------------- neighbour search
def knn_exact(points, queries, k, r2, chunk=2048):
    """Exact radius-limited top-k neighbour search — THE contract of lk_knn_query.

    d2 = (dx*dx + dy*dy) + dz*dz in fp32 (round after every op, no fma; the same
    formula the reference's tracker path uses, decoder.py:194-195).  Candidates
    are points with d2 <= r2; the k smallest by (d2, index) are returned in
    ascending order; empty slots hold idx = -1, d2 = FLT_MAX.  count = number of
    returned entries with d2 < r2 (strict), i.e. neural_point.py:1701-1706 applied
    to the returned list.  r2 is a scalar or a per-query [P] array (fp32).

    The reference uses approximate FAISS IVF (neural_point.py:1659-1708); parity
    with it is unpinned (oracle/__init__.py).
    """
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32).reshape(-1, 3))
    P, N = q.shape[0], pts.shape[0]
    r2 = np.broadcast_to(np.asarray(r2, dtype=np.float32).reshape(-1), (P,)) if np.ndim(r2) else \
        np.full((P,), np.float32(r2), dtype=np.float32)
    out_d = np.full((P, k), np.float32(FLT_MAX), dtype=np.float32)
    out_i = np.full((P, k), -1, dtype=np.int32)
    cnt = np.zeros((P,), dtype=np.int32)
    if N == 0 or P == 0:
        return out_d, out_i, cnt
    for s in range(0, P, chunk):
        qq = q[s:s + chunk]
        dx = qq[:, None, 0] - pts[None, :, 0]
        dy = qq[:, None, 1] - pts[None, :, 1]
        dz = qq[:, None, 2] - pts[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz                       # fp32, rounded per op
        rr = r2[s:s + chunk, None]
        d2m = np.where(d2 <= rr, d2, np.float32(np.inf))
        kk = min(k, N)
        # stable argsort on d2 keeps index order among equal distances => (d2, idx) order
        order = np.argsort(d2m, axis=1, kind='stable')[:, :kk]
        dsel = np.take_along_axis(d2m, order, axis=1)
        ok = np.isfinite(dsel)
        out_d[s:s + chunk, :kk] = np.where(ok, dsel, np.float32(FLT_MAX))
        out_i[s:s + chunk, :kk] = np.where(ok, order, -1).astype(np.int32)
        cnt[s:s + chunk] = (ok & (dsel < rr)).sum(1).astype(np.int32)
    return out_d, out_i, cnt


# --------------------------------------------------------------------------- interpolation
def softplus100(x):
    return F.softplus(x, beta=100)          # torch threshold: linear when 100*x > 20


def fourier(x, B, concat):
    """GaussianFourierFeatureTransform.forward (decoder.py:34-43).

    (2*pi*x) @ B with K=3: torch-CPU evaluates this as a0*b0 then two fused
    multiply-adds in k order (pinned by tests/test_oracle_golden.py::test_embed_fma_order);
    the HIP kernels use exactly that sequence so sin/cos see bit-identical arguments.
    """
    y = fourier_arg(x, B)
    return torch.cat((torch.sin(y), torch.cos(y)), dim=-1) if concat else torch.sin(y)


def fourier_arg(x, B):
    """(2 pi x) @ B for K = 3 in fp32 with the VALUES of the sequence a0*b0, fma(a1, b1, .), fma(a2, b2, .) on every host, and matmul's
    derivative.  On the host the goldens were captured on (and the build container) torch.matmul produces exactly that sequence - 0 of
    18.6 M arguments differ - but which kernel MKL picks depends on the CPU: on an AMD EPYC 9575F host (round 5, one kind of box of the
    GPU pool) its arguments differ from that sequence in the last bit, d/dp through 2 pi B cos(2 pi p B) with |B| ~ 25-32 amplifies an ulp
    a thousandfold, and the at-size tracker gradients of the kernels (which use the sequence) sat 7e-5 ... 1.2e-4 from this oracle instead of
    3e-6.  The fused multiply-add is evaluated in float64 (the product of two fp32 numbers is exact there) and rounded once; other dtypes
    (the float64 referee) go through matmul."""
    a = TWO_PI * x
    y = a @ B
    if y.dtype != torch.float32 or B.shape[0] != 3:
        return y
    with torch.no_grad():
        ad, Bd = a.detach().double(), B.detach().double()
        t = (a.detach()[..., 0:1] * B.detach()[0:1, :])                                  # a0 * b0, rounded to fp32
        t = (ad[..., 1:2] * Bd[1:2, :] + t.double()).float()                             # fma(a1, b1, t)
        t = (ad[..., 2:3] * Bd[2:3, :] + t.double()).float()                             # fma(a2, b2, t)
    return y + (t - y).detach()


def interp_weights(p, pos, idx, d2, r2, tracker):
    """Normalised inverse-squared-distance weights [P,k] (decoder.py:189-220, 440-472).

    tracker=True recomputes D differentiably from the positions and overwrites
    out-of-radius entries with 1e4 (decoder.py:191-198).
    """
    valid = idx >= 0
    I = idx.clamp(min=0).long()
    r2c = r2.reshape(-1, 1) if torch.is_tensor(r2) and r2.dim() > 0 else r2
    if tracker:
        D = torch.sum(torch.square(pos[I] - p.reshape(-1, 1, 3)), dim=-1)
        D = torch.where(valid, D, torch.full_like(D, FLT_MAX))
        D = torch.where(D > r2c, torch.full_like(D, 1e4), D)
    else:
        D = d2
    w = 1.0 / (D + 1e-10)
    w = torch.where((D > r2c) | (~valid), torch.zeros_like(w), w)
    w = w / torch.clamp(w.abs().sum(dim=1, keepdim=True), min=1e-12)
    return w, I


def interpolate(p, pos, feats, idx, d2, r2, count, min_nn, noise, tracker,
                relpos=None):
    """Feature at query points: c[P,C], has[P] (decoder.py:180-231 / 431-492).

    relpos = (B_r[3,10], W1[128,52], b1, W2[32,128], b2) enables the colour
    decoder's per-neighbour relative-position MLP (decoder.py:477-485).
    Rows with count < min_nn get the shared `noise` vector (decoder.py:228-229).
    """
    w, I = interp_weights(p, pos, idx, d2, r2, tracker)
    nf = feats[I]                                             # [P,k,C]
    if relpos is not None:
        B_r, W1, b1, W2, b2 = relpos
        rel = pos[I] - p.reshape(-1, 1, 3)
        emb = fourier(rel.reshape(-1, 3), B_r, concat=True).reshape(rel.shape[0], rel.shape[1], -1)
        x = torch.cat([emb, nf], dim=-1)
        nf = F.linear(softplus100(F.linear(x, W1, b1)), W2, b2)
    c = (w.unsqueeze(-1) * nf).sum(dim=1)
    has = count >= min_nn
    c = torch.where(has.reshape(-1, 1), c, noise.reshape(1, -1).expand_as(c))
    return c, has


# --------------------------------------------------------------------------- decoders
def _mlp5(e, c, W, prefix, act):
    """5-block trunk shared by both decoders (decoder.py:274-288, 523-533)."""
    h = e
    for i in range(5):
        h = F.linear(h, W[f'{prefix}.pts_linears.{i}.weight'], W[f'{prefix}.pts_linears.{i}.bias'])
        h = act(h)
        h = h + F.linear(c, W[f'{prefix}.fc_c.{i}.weight'], W[f'{prefix}.fc_c.{i}.bias'])
        if i == 2:
            h = torch.cat([e, h], -1)
    return F.linear(h, W[f'{prefix}.output_linear.weight'], W[f'{prefix}.output_linear.bias'])


def geo_mlp(p, c, W):
    """Occupancy logit per point (MLP_geometry.forward, decoder.py:263-288). relu trunk."""
    e = fourier(p, W['geo_decoder.embedder._B'], concat=False)
    return _mlp5(e, c, W, 'geo_decoder', F.relu).squeeze(-1)


def exposure_affine(W, exposure_feat):
    """MLP_exposure (decoder.py:326-342): 8 -> 128 softplus100 -> 12."""
    h = softplus100(F.linear(exposure_feat, W['color_decoder.mlp_exposure.linear1.weight'],
                             W['color_decoder.mlp_exposure.linear1.bias']))
    return F.linear(h, W['color_decoder.mlp_exposure.linear2.weight'],
                    W['color_decoder.mlp_exposure.linear2.bias'])


def color_mlp(p, c, W, affine=None, sigmoid=True):
    """RGB per point (MLP_color.forward, decoder.py:513-546). softplus(beta=100) trunk.

    affine (12,) = exposure transform applied before the sigmoid (decoder.py:534-540);
    sigmoid=False returns raw logits (decoder.py:541-542, mapper exposure path).
    """
    e = fourier(p, W['color_decoder.embedder._B'], concat=True)
    out = _mlp5(e, c, W, 'color_decoder', softplus100)
    if affine is not None:
        out = out @ affine[:9].reshape(3, 3) + affine[-3:]
    return torch.sigmoid(out) if sigmoid else out


# --------------------------------------------------------------------------- composite
def composite(occ, rgb, z, coef):
    """Alpha-composite S samples per ray (raw2outputs_nerf_color, common.py:382-422).

    occ [R,S], rgb [R,S,3], z [R,S] -> depth[R], var[R], color[R,3], w[R,S].
    """
    alpha = torch.sigmoid(coef * occ)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    wsum = w.sum(-1, keepdim=True) + 1e-10
    color = (w[..., None] * rgb).sum(-2) / wsum
    depth = (w * z).sum(-1) / wsum.squeeze(-1)
    tmp = z - depth.unsqueeze(-1)
    var = (w * tmp * tmp).sum(1)
    return depth, var, color, w


# --------------------------------------------------------------------------- full render
class RenderCfg:
    """Scalar knobs of the path (configs/point_slam.yaml + dataset yaml, resolved)."""

    def __init__(self, S=5, near_surface=0.98, far_surface=1.02, near_end=0.3, coef=0.1,
                 k=8, min_nn=2, radius_query=0.08, rel_pos=True, exposure=False):
        self.S, self.near_surface, self.far_surface, self.near_end = S, near_surface, far_surface, near_end
        self.coef, self.k, self.min_nn, self.radius_query = coef, k, min_nn, radius_query
        self.rel_pos, self.exposure = rel_pos, exposure


def relpos_params(W):
    return (W['color_decoder.embedder_rel_pos._B'],
            W['color_decoder.mlp_col_neighbor.linear1.weight'], W['color_decoder.mlp_col_neighbor.linear1.bias'],
            W['color_decoder.mlp_col_neighbor.linear2.weight'], W['color_decoder.mlp_col_neighbor.linear2.bias'])


def render_batch(cfg, rays_o, rays_d, gt_depth, pos, geo_feats, col_feats, W, stage,
                 tracker=False, r2_ray=None, noise_geo=None, noise_col=None, affine=None,
                 color_sigmoid=True, knn=None, near_pcl=False):
    """Renderer.render_batch_ray (Renderer.py:71-201) + NICER.forward (decoder.py:573-610).
    near_pcl: rendering.sample_near_pcl - the rays without a depth reading sample where the cloud is (Renderer.py:152-160) and the
    ones that find no cloud are masked out of valid_ray (Renderer.py:194-195).

    Returns dict(depth, var, color, valid_ray, has, z, idx, d2, count, w).
    r2_ray: per-ray squared query radius [R] (dynamic radius) or None (static).
    """
    R, S, C = rays_o.shape[0], cfg.S, geo_feats.shape[1]
    z, far = sample_z(gt_depth, cfg.near_surface, cfg.far_surface, cfg.near_end, S)
    not_near = None
    if near_pcl and bool((gt_depth.reshape(-1) <= 0).any()):
        zero = torch.nonzero(gt_depth.reshape(-1) <= 0).reshape(-1)
        z0, inv = sample_near_pcl(rays_o[zero].detach(), rays_d[zero].detach(), cfg.near_end, float(far), S, pos.detach().numpy(), cfg.radius_query)
        z = z.clone()
        z[zero] = torch.from_numpy(z0)
        not_near = zero[torch.from_numpy(inv)]
    p = sample_points(rays_o, rays_d, z)
    if r2_ray is None:
        r2 = torch.tensor(np.float32(cfg.radius_query ** 2))
        r2_np = np.float32(cfg.radius_query ** 2)
    else:
        r2 = r2_ray.float().reshape(-1, 1).repeat(1, S).reshape(-1)
        r2_np = r2.numpy()
    if knn is None:
        d2, idx, count = knn_exact(pos.detach().numpy(), p.detach().numpy(), cfg.k, r2_np)
    else:
        d2, idx, count = knn
    d2, idx, count = torch.from_numpy(np.asarray(d2)), torch.from_numpy(np.asarray(idx)), \
        torch.from_numpy(np.asarray(count))
    if noise_geo is None:
        noise_geo = torch.zeros(C)
    if noise_col is None:
        noise_col = torch.zeros(C)
    c_geo, has = interpolate(p, pos, geo_feats, idx, d2, r2, count, cfg.min_nn, noise_geo, tracker)
    occ = geo_mlp(p, c_geo, W)
    valid_ray = has.reshape(R, S).sum(1) >= int(S / 2 + 1)                 # decoder.py:259-260
    if stage == 'color':
        c_col, _ = interpolate(p, pos, col_feats, idx, d2, r2, count, cfg.min_nn, noise_col, tracker,
                               relpos=relpos_params(W) if cfg.rel_pos else None)
        rgb = color_mlp(p, c_col, W, affine=affine, sigmoid=color_sigmoid)
    else:
        rgb = torch.zeros(R * S, 3)
    # Renderer.py:184-186: occupancy of no-neighbour samples is overwritten with -100 by an
    # un-recorded in-place write; autograd still routes the gradient to the original logit.
    occ_eff = torch.where(has, occ, occ + (-100.0 - occ).detach())
    depth, var, color, w = composite(occ_eff.reshape(R, S), rgb.reshape(R, S, 3), z, cfg.coef)
    if not near_pcl:
        depth = torch.where(gt_depth.reshape(-1) > 0, depth, torch.zeros_like(depth))   # Renderer.py:197-198
    if not_near is not None:
        valid_ray = valid_ray.clone()
        valid_ray[not_near] = False                                                 # Renderer.py:194-195
    return dict(depth=depth, var=var, color=color, valid_ray=valid_ray, has=has, z=z, p=p,
                idx=idx, d2=d2, count=count, w=w, occ=occ, rgb=rgb)


# --------------------------------------------------------------------------- losses
def mapper_loss(depth, color, valid_ray, gt_depth, gt_color, stage, w_color):
    """Mapper.py:691-720 (no exposure): L1 depth (+ w_color * L1 colour in stage 'color')."""
    m = (gt_depth > 0) & valid_ray & (~torch.isnan(depth))
    geo = torch.abs(gt_depth[m] - depth[m]).sum()
    col = torch.abs(gt_color[m] - color[m]).sum() if stage == 'color' else torch.zeros(())
    return geo + w_color * col, geo, col, m


def tracker_loss(depth, var, color, gt_depth, gt_color, w_color, use_color=True, handle_dynamic=True):
    """Tracker.py:169-191: uncertainty-normalised L1 + colour L1; the outlier mask compares the normalised residual with 10 x its
    mean (handle_dynamic, the default of every config) or |gt - depth| with 10 x its median (Tracker.py:177-179)."""
    unc = var.detach()
    nan_mask = (~torch.isnan(depth)) & (~torch.isnan(unc))
    tmp = torch.abs(gt_depth - depth) / torch.sqrt(unc + 1e-10)
    if handle_dynamic:
        m = (tmp < 10 * tmp.mean()) & (gt_depth > 0) & nan_mask
    else:
        t2 = torch.abs(gt_depth - depth)
        m = (t2 < 10 * t2.median()) & (gt_depth > 0) & nan_mask
    geo = torch.clamp(tmp, min=0.0, max=1e3)[m].sum()
    col = torch.abs(gt_color - color)[m].sum()
    loss = geo + w_color * col if use_color else geo
    return loss, geo, col, m


# --------------------------------------------------------------------------- Adam
def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (amsgrad=False, weight_decay=0), in place.

    Mapper.py:570,723; Tracker.py:352,194.  `step` is the 1-based step count.
    """
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
    return p


def knn_tree(points, queries, k, r2, tree=None):
    """Same contract as knn_exact for large clouds (CPU baseline timing): a KD-tree proposes the k nearest
    within the radius, distances are then recomputed in fp32 with the contract's formula and re-ordered by
    (d2, index).  The tree proposes 2k candidates, of which the k smallest by the fp32 key are kept: fp64 and fp32 orderings disagree
    around the k-th neighbour wherever distances tie to rounding (points inserted at fixed ratios along the rays that are sampled at
    the same ratios tie exactly) - with k proposals 3 % of the lists of a small synthetic scene differed from knn_exact, with 2k none
    (only differs from knn_exact if more than k candidates tie with the k-th).
    tree: a cKDTree already built over `points` as float64 (a caller that searches one cloud many times)."""
    from scipy.spatial import cKDTree
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32).reshape(-1, 3))
    P = q.shape[0]
    r2a = np.broadcast_to(np.asarray(r2, dtype=np.float32).reshape(-1), (P,)) if np.ndim(r2) else np.full((P,), np.float32(r2), np.float32)
    if tree is None:
        tree = cKDTree(pts.astype(np.float64))
    rmax = float(np.sqrt(r2a.max())) * 1.001
    _, ii = tree.query(q.astype(np.float64), k=2 * k, distance_upper_bound=rmax, workers=-1)
    ok = ii < pts.shape[0]
    I = np.where(ok, ii, 0)
    d = pts[I] - q[:, None, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    ok &= d2 <= r2a[:, None]
    d2 = np.where(ok, d2, np.float32(np.inf)).astype(np.float32)
    key = np.lexsort((np.where(ok, ii, np.iinfo(np.int64).max), d2), axis=1)
    d2 = np.take_along_axis(d2, key, 1)[:, :k]
    idx = np.take_along_axis(np.where(ok, ii, -1), key, 1)[:, :k].astype(np.int32)
    okk = np.isfinite(d2)
    cnt = (okk & (d2 < r2a[:, None])).sum(1).astype(np.int32)
    return np.where(okk, d2, np.float32(FLT_MAX)).astype(np.float32), idx, cnt


# ------------------------------------------------------------------ map maintenance around the hot loop
def remap_linear_zero(img, u, v):
    """cv2.remap(img, u, v, INTER_LINEAR) with the default BORDER_CONSTANT(0), float32 image and maps, restated from
    OpenCV's documented algorithm (imgproc remap: the sub-pixel position is quantised to 1/32 - INTER_TAB_SIZE = 32,
    cvRound = round-half-even - and the four taps are combined with the float32 table weights
    [(1-ay)(1-ax), (1-ay)ax, ay(1-ax), ay ax] in that order; taps outside the image read 0).
    PARITY UNPINNED: cv2 is not installed in the build image, the reference's tests never touch this call."""
    img = np.asarray(img, np.float32)
    H, W = img.shape
    u = np.asarray(u, np.float32)
    v = np.asarray(v, np.float32)
    big = (np.abs(u) > 1e7) | (np.abs(v) > 1e7) | ~np.isfinite(u) | ~np.isfinite(v)
    us, vs = np.where(big, np.float32(-1e6), u), np.where(big, np.float32(-1e6), v)
    sx = np.rint(us.astype(np.float64) * 32.0).astype(np.int64)
    sy = np.rint(vs.astype(np.float64) * 32.0).astype(np.int64)
    ix, iy, ax, ay = sx >> 5, sy >> 5, (sx & 31).astype(np.float32) / np.float32(32), (sy & 31).astype(np.float32) / np.float32(32)
    one = np.float32(1)

    def tap(y, x):
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        return np.where(ok, img[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)], np.float32(0))
    w00, w01, w10, w11 = (one - ay) * (one - ax), (one - ay) * ax, ay * (one - ax), ay * ax
    out = tap(iy, ix) * w00
    out = out + tap(iy, ix + 1) * w01
    out = out + tap(iy + 1, ix) * w10
    out = out + tap(iy + 1, ix + 1) * w11
    return out.astype(np.float32)


def frustum_rows(pos, c2w, depth, fx, fy, cx, cy, H, W, edge):
    """Mapper.get_mask_from_c2w (src/Mapper.py:165-217): indices of the cloud points inside the (edge-cropped) image
    of pose c2w and not behind the observed surface by more than 0.5 m.  Arithmetic as the reference: w2c =
    inv(c2w) in float32, projection in float64 (points are python floats there), uv cast to float32, bilinear depth
    lookup, zero depths replaced by the max of the sampled depths.  Sums are written out in a fixed order
    ((a+b)+c)+d so the HIP kernel can match bit for bit."""
    pos = np.asarray(pos, np.float32).astype(np.float64)
    w2c = np.linalg.inv(np.asarray(c2w, np.float32)).astype(np.float32).astype(np.float64)
    x, y, z = pos[:, 0], pos[:, 1], pos[:, 2]
    cam = [((w2c[r, 0] * x + w2c[r, 1] * y) + w2c[r, 2] * z) + w2c[r, 3] for r in range(3)]
    cx_, cy_, cz_ = -cam[0], cam[1], cam[2]                       # cam_cord[:, 0] *= -1
    zz = cz_ + 1e-5
    u = ((fx * cx_ + cx * cz_) / zz).astype(np.float32)
    v = ((fy * cy_ + cy * cz_) / zz).astype(np.float32)
    d = remap_linear_zero(depth, u, v)
    if d.size:
        d = np.where(d == 0, d.max(), d)
    m = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge)
    m = m & (0 <= -zz) & (-zz <= (d + np.float32(0.5)).astype(np.float64))
    return np.nonzero(m)[0].astype(np.int32)


def add_points(rays_o, rays_d, gt_depth, cloud_pos, r2_add, near_surface, far_surface, n_add=3):
    """NeuralPointCloud.add_neural_points, geometry only (src/neural_point.py:1557-1631): rays with depth > 0 whose
    surface point o + d*depth has NO cloud point with squared distance < r2_add (scalar or per-ray) each contribute
    n_add points at z = linspace(near*depth, far*depth).  Rays of the same call are not de-duplicated against each
    other.  Returns (accepted ray indices, new points [3*k, 3]).  d2 as in knn_exact: (dx*dx + dy*dy) + dz*dz in f32."""
    ro, rd, gd = (torch.as_tensor(t, dtype=torch.float32) for t in (rays_o, rays_d, gt_depth))
    keep = torch.nonzero(gd > 0).reshape(-1)
    p = (ro[keep] + rd[keep] * gd[keep, None]).numpy()
    r2 = np.broadcast_to(np.asarray(r2_add, np.float32), gd.shape)[keep.numpy()] if np.ndim(r2_add) else np.float32(r2_add)
    cloud = np.asarray(cloud_pos, np.float32).reshape(-1, 3)
    if cloud.shape[0]:
        d2, idx, cnt = knn_exact(cloud, p, 8, r2)
        ok = cnt == 0
    else:
        ok = np.ones(p.shape[0], bool)
    acc = keep[torch.from_numpy(ok)]
    t = torch.linspace(0., 1., n_add)
    dsurf = gd[acc, None].repeat(1, n_add)
    z = near_surface * dsurf * (1. - t) + far_surface * dsurf * t
    pts = ro[acc, None, :] + rd[acc, None, :] * z[..., None]
    return acc.to(torch.int32), pts.reshape(-1, 3)


def color_grad_mag(color):
    """|Sobel| of the grey image as the reference computes it (Tracker.py:244-248, common.py:182-185, 209-212) with
    scikit-image 0.19.3 (env.yaml:134; not installed here - restated from its source, the convolution itself is
    scipy.ndimage.convolve which IS installed).  The reference's colour images are float64 (uint8 / 255. in numpy,
    datasets.py), so the whole pre-pass runs in float64: rgb2gray = rgb @ [0.2125, 0.7154, 0.0721],
    sobel_h / sobel_v = convolve(grey, [1,0,-1] x [1,2,1]/4, mode='reflect'), magnitude sqrt(gx^2 + gy^2).
    Here the image arrives as float32 and is promoted."""
    import scipy.ndimage as ndi
    rgb = np.asarray(color, np.float32).astype(np.float64)
    grey = (rgb[..., 0] * 0.2125 + rgb[..., 1] * 0.7154) + rgb[..., 2] * 0.0721
    edge, smooth = np.array([1.0, 0.0, -1.0]), np.array([1.0, 2.0, 1.0]) / 4.0
    gy = ndi.convolve(grey, edge.reshape(3, 1) * smooth.reshape(1, 3), mode='reflect')      # sobel_h: edges along axis 0
    gx = ndi.convolve(grey, smooth.reshape(3, 1) * edge.reshape(1, 3), mode='reflect')      # sobel_v
    return np.sqrt(gx ** 2 + gy ** 2)


def radius_maps(color, radius_add_max, radius_add_min, ratio, thr):
    """Per-pixel dynamic radii (Tracker.py:243-258; Mapper.py:854-872 does the same): gradient magnitude clipped to
    [0, thr], piecewise-linear map [0, 0.01, thr] -> [r_max, r_max, r_min] (scipy interp1d arithmetic
    slope * (x - x_lo) + y_lo), all float64; r_query = ratio * r_add.  Returns (grad_mag, r_add, r_query) float64."""
    g = color_grad_mag(color)
    x = np.clip(g, 0.0, thr)

    def interp(y_max, y_min):
        slope = (y_min - y_max) / (thr - 0.01)
        return np.where(x <= 0.01, y_max, slope * (x - 0.01) + y_max)
    return g, interp(radius_add_max, radius_add_min), interp(ratio * radius_add_max, ratio * radius_add_min)


def top_grad_pixels(grad_mag, k, window, depth=None, depth_limit=False):
    """Pixel pool of get_selected_index_with_grad (common.py:198-234): the k = ratio*n pixels with the largest gradient
    magnitude of the WHOLE image, then restricted to the window [H0,H1) x [W0,W1) and to depth > 0 (<= 5 with
    depth_limit).  np.argpartition leaves the choice among equal magnitudes at the cut unspecified; the contract here
    is: every pixel strictly above the k-th largest value, then ties in ascending flat index.  Returns sorted flat
    indices (int32)."""
    g = np.asarray(grad_mag, np.float32)
    H, W = g.shape
    flat = g.reshape(-1)
    k = min(int(k), flat.size)
    if k <= 0:
        return np.zeros(0, np.int32)
    kth = np.sort(flat)[flat.size - k]
    above = np.nonzero(flat > kth)[0]
    ties = np.nonzero(flat == kth)[0][:k - above.size]
    sel = np.sort(np.concatenate([above, ties]))
    ih, iw = sel // W, sel % W
    H0, H1, W0, W1 = window
    m = (ih >= H0) & (ih < H1) & (iw >= W0) & (iw < W1)
    if depth is not None:
        d = np.asarray(depth, np.float32).reshape(-1)[sel]
        m &= (d > 0) & ((d <= 5.0) if depth_limit else True)
    return sel[m].astype(np.int32)


def sample_near_pcl(rays_o, rays_d, near, far, num, cloud_pos, radius_query, intervals=25):
    """NeuralPointCloud.sample_near_pcl (src/neural_point.py:1734-1786): probe `intervals` depths between near and far on
    each ray, keep the span between the first and the last probe that has a cloud point within radius_query
    (count > 0), and place `num` samples evenly in it (numpy float64 linspace, cast to float32).  Rays with fewer than two
    supported probes are 'invalid' and keep linspace(near, far, num).  Returns (z [n, num] f32, invalid [n] bool)."""
    ro = torch.as_tensor(rays_o, dtype=torch.float32).reshape(-1, 3)
    rd = torch.as_tensor(rays_d, dtype=torch.float32).reshape(-1, 3)
    n = rd.shape[0]
    far = float(far)
    zp = torch.linspace(near, far, steps=intervals)
    pts = (ro[:, None, :] + rd[:, None, :] * zp[None, :, None]).reshape(-1, 3)
    _, _, cnt = knn_exact(np.asarray(cloud_pos, np.float32), pts.numpy(), 8, np.float32(radius_query ** 2))
    sup = cnt.reshape(n, intervals) > 0
    invalid = sup.sum(-1) < 2
    sect = np.linspace(near, far, intervals)
    z = np.tile(np.linspace(near, far, num), (n, 1))
    for r in np.nonzero(~invalid)[0]:
        c = np.nonzero(sup[r])[0]
        z[r] = np.linspace(sect[c[0]], sect[c[1]], num=num)          # item[0], item[1]: the first TWO supported probes
    return z.astype(np.float32), invalid


# ------------------------------------------------------------------ point-insertion schedule of a mapped frame
def filter_point_before_add(rays_o, rays_d, gt_depth, prev_c2w, fx, fy, cx, cy, H, W):
    """Mapper.filter_point_before_add (src/Mapper.py:137-163): True for rays whose surface point falls OUTSIDE the image of
    the previously mapped pose.  float32 world->camera product, float64 intrinsics product, uv cast to float32."""
    pts = (torch.as_tensor(rays_o) + torch.as_tensor(rays_d) * torch.as_tensor(gt_depth)[:, None]).reshape(-1, 3).numpy().astype(np.float32)
    w2c = np.linalg.inv(np.asarray(prev_c2w, np.float32))
    homo = np.concatenate([pts, np.ones_like(pts[:, :1])], axis=1).reshape(-1, 4, 1)
    cam = (w2c @ homo)[:, :3]
    K = np.array([[fx, .0, cx], [.0, fy, cy], [.0, .0, 1.0]])
    cam[:, 0] *= -1
    uv = K @ cam
    z = uv[:, -1:] + 1e-5
    uv = (uv[:, :2] / z).astype(np.float32)[..., 0]
    inside = (uv[:, 0] < W) & (uv[:, 0] > 0) & (uv[:, 1] < H) & (uv[:, 1] > 0)
    return torch.from_numpy(~inside)


def first_frame_add_count(gt_depth_img, pixels_adding):
    """idx == 0: clamp(pixels_adding * (median(depth) / 2.5)^2, pixels_adding, 3 pixels_adding) (Mapper.py:421-425);
    torch.median = lower median over the whole image, zeros included."""
    d = torch.as_tensor(gt_depth_img, dtype=torch.float32)
    return int(torch.clamp(pixels_adding * ((d.median() / 2.5) ** 2), min=pixels_adding, max=pixels_adding * 3).int().item())


def mapping_iterations(num_joint_iters, frame_pts_add, min_iter_ratio):
    """idx > 0: clip(int(iters * added / 300), int(min_iter_ratio * iters), 2 iters) (Mapper.py:572-574)."""
    return int(np.clip(int(num_joint_iters * frame_pts_add / 300), int(min_iter_ratio * num_joint_iters), 2 * num_joint_iters))


def add_points_schedule(idx, depth_img, color_img, c2w, prev_c2w, cloud_pos, intr, cfg, draws, r_add_map=None):
    """The insertion passes of one mapped frame (src/Mapper.py:421-482) on explicit pixel draws:
      idx == 0 (or filter_before_add_points off): ONE pass over `draws['main']` (first_frame_add_count of them for idx == 0);
      idx > 0: the main draws restricted to surface points OUTSIDE the previous view, then `draws['overlap']` (1000 pixels)
               restricted to points INSIDE it - holes in already seen areas get refilled;
      pixels_based_on_color_grad > 0: `draws['grad']` (flat indices of high-gradient pixels, sorted) with is_pts_grad
               (radius_min instead of radius_add unless a dynamic radius map is given).
    Every pass tests against the cloud INCLUDING the points of the passes before it (add_neural_points re-indexes).
    cfg: dict(pixels_adding, pixels_grad, radius_add, radius_min, near, far, filter_before).  draws are flat indices into
    the H x W image.  Returns (frame_pts_add, per-pass accepted counts, grown cloud [N', 3])."""
    fx, fy, cx, cy = intr
    depth = torch.as_tensor(depth_img, dtype=torch.float32)
    Hh, Ww = depth.shape
    cloud = torch.as_tensor(cloud_pos, dtype=torch.float32).reshape(-1, 3)
    counts = []

    def rays(flat):
        flat = torch.as_tensor(flat).long()
        i, j = (flat % Ww).float(), torch.div(flat, Ww, rounding_mode='floor').float()
        ro, rd = rays_from_uv(i, j, torch.as_tensor(c2w, dtype=torch.float32), fx, fy, cx, cy)
        gd = depth.reshape(-1)[flat]
        keep = gd > 0                                          # get_samples(depth_filter=True)
        r = None if r_add_map is None else torch.as_tensor(r_add_map).reshape(-1)[flat][keep]
        return ro[keep], rd[keep], gd[keep], r

    def insert(ro, rd, gd, r, pts_grad):
        nonlocal cloud
        if ro.shape[0] == 0:
            counts.append(0)
            return
        if r is not None:
            r2 = (r.double() ** 2).float().numpy()
        else:
            r2 = np.float32((cfg['radius_min'] if pts_grad else cfg['radius_add']) ** 2)
        acc, pts = add_points(ro, rd, gd, cloud.numpy(), r2, cfg['near'], cfg['far'])
        cloud = torch.cat([cloud, pts])
        counts.append(int(acc.shape[0]))

    n_main = first_frame_add_count(depth, cfg['pixels_adding']) if idx == 0 else cfg['pixels_adding']
    ro, rd, gd, r = rays(draws['main'][:n_main])
    if cfg['filter_before'] and idx != 0:
        out = filter_point_before_add(ro, rd, gd, prev_c2w, fx, fy, cx, cy, Hh, Ww)
        insert(ro[out], rd[out], gd[out], None if r is None else r[out], False)
        ro, rd, gd, r = rays(draws['overlap'])
        out = filter_point_before_add(ro, rd, gd, prev_c2w, fx, fy, cx, cy, Hh, Ww)
        insert(ro[~out], rd[~out], gd[~out], None if r is None else r[~out], False)
    else:
        insert(ro, rd, gd, r, False)
    if cfg['pixels_grad'] > 0:
        ro, rd, gd, r = rays(draws['grad'])
        insert(ro, rd, gd, r, True)
    return sum(counts), counts, cloud


def keyframe_overlap_fractions(vertices, est_c2ws, fx, fy, cx, cy, H, W):
    """percent_inside of src/Mapper.py:250-270 for every keyframe pose, statement by statement in numpy: w2c = inv(c2w), homogeneous
    product, K @ cam (x NOT mirrored - the flip is commented out in this function of the reference), z = uv_z + 1e-5, float32 uv,
    20-pixel edge, z < 0."""
    vertices = np.asarray(vertices, dtype=np.float32)
    out = []
    for c2w in est_c2ws:
        w2c = np.linalg.inv(np.asarray(c2w, dtype=np.float32))
        ones = np.ones_like(vertices[:, 0]).reshape(-1, 1)
        homo = np.concatenate([vertices, ones], axis=1).reshape(-1, 4, 1)
        cam = (w2c @ homo)[:, :3]
        K = np.array([[fx, .0, cx], [.0, fy, cy], [.0, .0, 1.0]]).reshape(3, 3)
        uv = K @ cam
        z = uv[:, -1:] + 1e-5
        uv = (uv[:, :2] / z).astype(np.float32)
        edge = 20
        mask = (uv[:, 0] < W - edge) * (uv[:, 0] > edge) * (uv[:, 1] < H - edge) * (uv[:, 1] > edge)
        mask = mask & (z[:, :, 0] < 0)
        out.append(mask.reshape(-1).sum() / uv.shape[0])
    return np.asarray(out)
