"""Pose / ray helpers with the reference's names and conventions (src/common.py:104-138, 237-259, 301-379,
425-442).  These are the per-frame (not per-iteration) host-side pieces: small torch expressions on whatever
device the inputs live on.  Inside the optimisation loops the same arithmetic runs in lk_rays_from_pose /
lk_gather_rays / lk_pose_bwd."""
import numpy as np
import torch


def quad2rotation(quad):
    """(B,4) un-normalised quaternions (w,x,y,z) -> (B,3,3).  Differentiable (src/common.py:301-324)."""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    r = torch.stack([
        1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
        two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
        two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)], dim=-1)
    return r.reshape(-1, 3, 3)


def get_camera_from_tensor(inputs):
    """(7,) or (B,7) [quat wxyz, T] -> 3x4 / Bx3x4 (src/common.py:327-343)."""
    single = inputs.dim() == 1
    x = inputs.unsqueeze(0) if single else inputs
    RT = torch.cat([quad2rotation(x[:, :4]), x[:, 4:, None]], 2)
    return RT[0] if single else RT


def get_tensor_from_camera(RT, Tquad=False):
    """4x4 / 3x4 matrix -> float32 [quat wxyz, T] on the matrix's device (src/common.py:354-379)."""
    from scipy.spatial.transform import Rotation
    dev = RT.device if torch.is_tensor(RT) else 'cpu'
    M = RT.detach().cpu().numpy() if torch.is_tensor(RT) else np.asarray(RT)
    quad = np.roll(Rotation.from_matrix(M[:3, :3].astype(np.float64)).as_quat(), 1)
    T = M[:3, 3]
    v = np.concatenate([T, quad]) if Tquad else np.concatenate([quad, T])
    return torch.from_numpy(v).float().to(dev)


def get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, device=None):
    """src/common.py:104-120: i = column, j = row (flattened); un-normalised directions."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w).to(i.device)
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def get_rays(H, W, fx, fy, cx, cy, c2w, device, crop_edge=0):
    """All rays of an image, (H', W', 3) each (src/common.py:425-442)."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device)
    jj, ii = torch.meshgrid(torch.linspace(crop_edge, H - 1 - crop_edge, H - 2 * crop_edge, device=device),
                            torch.linspace(crop_edge, W - 1 - crop_edge, W - 2 * crop_edge, device=device), indexing='ij')
    ro, rd = get_rays_from_uv(ii.reshape(-1), jj.reshape(-1), c2w, H, W, fx, fy, cx, cy)
    shp = (H - 2 * crop_edge, W - 2 * crop_edge, 3)
    return ro.reshape(shp), rd.reshape(shp)


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, device,
                depth_filter=False, return_index=False, depth_limit=None, generator=None):
    """n rays drawn with replacement from the window (src/common.py:160-172, 237-259)."""
    h, w = H1 - H0, W1 - W0
    idx = torch.randint(h * w, (n,), device=device, generator=generator)
    i = (W0 + idx % w).float()
    j = (H0 + torch.div(idx, w, rounding_mode='floor')).float()
    sd = depth[j.long(), i.long()]
    sc = color[j.long(), i.long()]
    ro, rd = get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy)
    if depth_filter:
        m = sd > 0
        if depth_limit is not None:
            m = m & (sd < depth_limit)
        ro, rd, sd, sc, i, j = ro[m], rd[m], sd[m], sc[m], i[m], j[m]
    if return_index:
        return ro, rd, sd, sc, i.to(torch.int64), j.to(torch.int64)
    return ro, rd, sd, sc
