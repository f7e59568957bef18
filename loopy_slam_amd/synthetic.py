"""Synthetic RGB-D room for benchmarks and parity tests at full size.

Scene: axis-aligned box room 6 x 4 x 3 m (walls get a sinusoidal relief), camera inside on a
closed loop, intrinsics of the TUM config (fx 517.3, fy 516.5, cx 318.6, cy 255.3, 640x480).
Depth = exact ray/box hit (metres, 2 % zero holes on request), RGB = procedural texture in [0,1].
The neural point cloud is laid down the way NeuralPointCloud.add_neural_points does
(src/neural_point.py:1557-1631): three points per chosen pixel at (0.98, 1.0, 1.02) * depth,
features ~ N(0, 0.1).  Everything is plain torch on whatever device is asked for; no dataset,
no network.
"""
import math

import torch

ROOM = (6.0, 4.0, 3.0)
TUM_INTR = dict(H=480, W=640, fx=517.3, fy=516.5, cx=318.6, cy=255.3)


def loop_pose(k, n=200, device='cpu'):
    """c2w [4,4] of pose k of n on a closed loop inside the room (camera looks down -z)."""
    a = 2 * math.pi * k / n
    eye = torch.tensor([0.9 * math.cos(a), 0.6 * math.sin(a), 0.1 * math.sin(2 * a)])
    yaw = a + math.pi / 2
    fwd = torch.tensor([math.cos(yaw), math.sin(yaw), -0.1])
    fwd = fwd / fwd.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    up2 = torch.linalg.cross(right, fwd)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up2, -fwd, eye
    return c2w.to(device)


def handheld_pose(k, n=600, device='cpu'):
    """Pose k of a hand-held walk along the loop: loop_pose(k, n) (n = 600: 0.6 degrees of yaw and 0.6-0.9 cm per frame) plus a
    smooth sway of the camera centre (sinusoids of 7-13 frame periods, 1-1.5 cm) and of its orientation (0.5-0.6 degrees about two
    camera axes) - per-frame motion of 1-2 cm and 0.5-1.2 degrees like a Replica / TUM sequence, and a velocity that CHANGES from frame
    to frame: a constant-velocity prediction from the two previous true poses misses the pose by about 1 cm and 0.4 degrees, which
    is what the tracker has to remove (tests/oracle_slam.py: prior_only_metrics)."""
    c2w = loop_pose(k, n, 'cpu').double()
    t = float(k)
    sway = torch.tensor([0.012 * math.sin(2 * math.pi * t / 11.0 + 0.3) + 0.008 * math.sin(2 * math.pi * t / 7.0 + 1.1),
                         0.010 * math.sin(2 * math.pi * t / 13.0 + 2.0) + 0.008 * math.cos(2 * math.pi * t / 7.5),
                         0.015 * math.sin(2 * math.pi * t / 9.0 + 0.7)], dtype=torch.float64)
    a = math.radians(0.6) * math.sin(2 * math.pi * t / 9.0 + 1.3)          # about the camera's x axis (pitch)
    b = math.radians(0.5) * math.sin(2 * math.pi * t / 8.0 + 0.2)          # about the camera's y axis (yaw)
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]], dtype=torch.float64)
    Ry = torch.tensor([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]], dtype=torch.float64)
    out = c2w.clone()
    out[:3, :3] = c2w[:3, :3] @ Ry @ Rx
    out[:3, 3] = c2w[:3, 3] + sway
    return out.float().to(device)


def sequence_pose(k, motion='loop', n_poses=200, device='cpu'):
    """Pose k of a synthetic sequence: 'loop' = loop_pose(k, n_poses), 'handheld' = handheld_pose(k) (data.motion of the config)."""
    if motion == 'handheld':
        return handheld_pose(k, device=device)
    return loop_pose(k, n_poses, device)


def pixel_rays(c2w, i, j, intr=TUM_INTR):
    """get_rays_from_uv convention (src/common.py:104-120): i = column, j = row, un-normalised d."""
    dirs = torch.stack([(i - intr['cx']) / intr['fx'], -(j - intr['cy']) / intr['fy'], -torch.ones_like(i)], -1)
    rays_d = (dirs[:, None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, 3].expand_as(rays_d)
    return rays_o.contiguous(), rays_d.contiguous()


def room_depth(rays_o, rays_d):
    """Distance parameter t (in units of |d|, i.e. the reference's z-depth) to the box walls + relief."""
    half = torch.tensor(ROOM, device=rays_o.device) / 2
    inv = 1.0 / torch.where(rays_d.abs() < 1e-9, torch.full_like(rays_d, 1e-9), rays_d)
    t1 = (half - rays_o) * inv
    t2 = (-half - rays_o) * inv
    t = torch.maximum(t1, t2).min(dim=-1).values
    hit = rays_o + rays_d * t[:, None]
    relief = 0.03 * torch.sin(3.0 * hit[:, 0]) * torch.sin(2.5 * hit[:, 1] + 1.0) * torch.cos(2.0 * hit[:, 2])
    return (t * (1.0 + relief / t.clamp(min=0.5))).float()


def room_color(points):
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    r = 0.5 + 0.5 * torch.sin(2.1 * x + 0.5) * torch.cos(1.7 * y)
    g = 0.5 + 0.5 * torch.sin(1.3 * y + 2.0 * z)
    b = 0.5 + 0.5 * torch.cos(2.9 * z + 0.7 * x)
    return torch.stack([r, g, b], -1).clamp(0, 1).float()


# 'furnished' scene (data.scene of the config; the accuracy runs): the plain room is one or two smooth walls per view - a camera can slide
# along them and only a 3 cm relief of 2 m period and a texture of 3 m period say no, so pose tracking is ill-conditioned by construction.
# Axis-aligned boxes along the walls (cabinets, shelves, a table: depth edges and faces in three orientations within every view), a finer
# relief and a finer texture make it a scene a tracker can lock on to, like the furniture of a Replica room.  (centre, half size) in metres;
# everything stays outside the camera's walk (an ellipse of 0.9 x 0.6 m around the room's centre).
FURNITURE = (
    ((2.60, 0.00, -0.90), (0.40, 0.90, 0.60)), ((2.70, -1.50, 0.30), (0.30, 0.35, 1.20)), ((2.55, 1.45, -0.50), (0.45, 0.40, 1.00)),
    ((-2.60, 0.30, -0.80), (0.40, 1.10, 0.70)), ((-2.70, -1.40, 0.20), (0.30, 0.40, 1.30)), ((-2.65, 1.60, 0.60), (0.35, 0.30, 0.50)),
    ((0.20, 1.70, -0.85), (1.00, 0.30, 0.65)), ((-1.50, 1.75, 0.40), (0.35, 0.25, 0.90)), ((1.60, 1.70, 0.55), (0.40, 0.30, 0.45)),
    ((-0.30, -1.70, -0.90), (0.90, 0.30, 0.60)), ((1.50, -1.75, 0.30), (0.35, 0.25, 1.00)), ((-1.70, -1.70, 0.50), (0.30, 0.30, 0.60)),
    ((1.75, 0.95, -1.20), (0.35, 0.35, 0.30)), ((-1.80, -0.90, -1.15), (0.30, 0.40, 0.35)),
)


def furnished_hit(rays_o, rays_d):
    """(t [R], object [R]) of the furnished room: object -1 = the walls (room_depth with a second, finer relief), k >= 0 = box k of
    FURNITURE (slab test per box, the nearest entry in front of the camera)."""
    half = torch.tensor(ROOM, device=rays_o.device) / 2
    inv = 1.0 / torch.where(rays_d.abs() < 1e-9, torch.full_like(rays_d, 1e-9), rays_d)
    t = torch.maximum((half - rays_o) * inv, (-half - rays_o) * inv).min(dim=-1).values
    hit = rays_o + rays_d * t[:, None]
    relief = 0.03 * torch.sin(3.0 * hit[:, 0]) * torch.sin(2.5 * hit[:, 1] + 1.0) * torch.cos(2.0 * hit[:, 2]) + \
        0.025 * torch.sin(11.0 * hit[:, 0] + 0.5) * torch.cos(9.0 * hit[:, 1]) * torch.sin(10.0 * hit[:, 2] + 0.3)
    t = t * (1.0 + relief / t.clamp(min=0.5))
    obj = torch.full_like(t, -1.0)
    c = torch.tensor([b[0] for b in FURNITURE], device=rays_o.device)         # [K,3]
    h = torch.tensor([b[1] for b in FURNITURE], device=rays_o.device)
    lo = (c - h)[None] - rays_o[:, None, :]                                    # [R,K,3]
    hi = (c + h)[None] - rays_o[:, None, :]
    t1, t2 = lo * inv[:, None, :], hi * inv[:, None, :]
    tn = torch.minimum(t1, t2).max(dim=-1).values                              # entry
    tf = torch.maximum(t1, t2).min(dim=-1).values                              # exit
    ok = (tn < tf) & (tn > 1e-3)
    tn = torch.where(ok, tn, torch.full_like(tn, float('inf')))
    tb, kb = tn.min(dim=1)
    closer = tb < t
    return torch.where(closer, tb, t).float(), torch.where(closer, kb.float(), obj)


def furnished_color(points, obj):
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    base = room_color(points)
    fine = torch.stack([0.5 + 0.5 * torch.sin(13.0 * x + 2.0 * y) * torch.cos(11.0 * z),
                        0.5 + 0.5 * torch.sin(12.0 * y - 3.0 * z + 1.0),
                        0.5 + 0.5 * torch.cos(14.0 * z + 9.0 * x) * torch.sin(10.0 * y + 0.4)], -1)
    k = obj.clamp(min=0)
    tint = torch.stack([0.5 + 0.5 * torch.sin(1.7 * k + 0.3), 0.5 + 0.5 * torch.sin(2.3 * k + 1.9), 0.5 + 0.5 * torch.sin(3.1 * k + 4.0)], -1)
    wall = 0.65 * base + 0.35 * fine
    box = 0.5 * tint + 0.2 * base + 0.3 * fine
    return torch.where((obj >= 0)[:, None], box, wall).clamp(0, 1).float()


def render_frame(k, intr=TUM_INTR, holes=0.02, device='cpu', seed=1219, n_poses=200, motion='loop', scene='plain'):
    """Synthetic RGB-D frame k: (depth [H,W], color [H,W,3], c2w [4,4])."""
    H, W = intr['H'], intr['W']
    c2w = sequence_pose(k, motion, n_poses, device)
    jj, ii = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing='ij')
    ro, rd = pixel_rays(c2w, ii.reshape(-1), jj.reshape(-1), intr)
    if scene == 'furnished':
        d, obj = furnished_hit(ro, rd)
        col = furnished_color(ro + rd * d[:, None], obj)
    else:
        d = room_depth(ro, rd)
        col = room_color(ro + rd * d[:, None])
    if holes > 0:
        g = torch.Generator(device='cpu').manual_seed(seed + k)
        m = (torch.rand(H * W, generator=g) < holes).to(device)
        d = torch.where(m, torch.zeros_like(d), d)
    return d.reshape(H, W), col.reshape(H, W, 3), c2w


def build_cloud(n_points, device='cpu', seed=1219, intr=TUM_INTR, n_views=24, c_dim=32):
    """Point cloud of about n_points (multiple of 3): pixels of n_views loop poses, 3 points per
    pixel along the ray at (0.98, 1.0, 1.02) * depth; geo/col features ~ N(0, 0.1)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    per_view = (n_points // 3 + n_views - 1) // n_views
    pts = []
    for v in range(n_views):
        c2w = loop_pose(v * (200 // n_views), 200, 'cpu')
        i = torch.rand(per_view, generator=g) * (intr['W'] - 1)
        j = torch.rand(per_view, generator=g) * (intr['H'] - 1)
        ro, rd = pixel_rays(c2w, i, j, intr)
        d = room_depth(ro, rd)
        for t in (0.98, 1.0, 1.02):
            pts.append(ro + rd * (d * t)[:, None])
    pos = torch.cat(pts, 0)[:n_points].float().contiguous()
    geo = (0.1 * torch.randn(pos.shape[0], c_dim, generator=g)).float()
    col = (0.1 * torch.randn(pos.shape[0], c_dim, generator=g)).float()
    return pos.to(device), geo.to(device), col.to(device)


ROOM_PITCH = 8.0          # metres between the rooms of a multi-room map (x axis): the walls of neighbouring rooms are 2 m apart


def room_offset(room, device='cpu'):
    return torch.tensor([ROOM_PITCH * room, 0.0, 0.0], device=device)


def pose_in_room(c2w, room):
    """The pose moved into room `room` of a multi-room map: every room is the same closed box, so the frame it sees is the base frame."""
    out = c2w.clone()
    out[:3, 3] += room_offset(room, c2w.device)
    return out


def build_cloud_online(eng, n_points, seed=1219, intr=TUM_INTR, c_dim=32, pixels_per_call=6000, points_per_room=100_000):
    """The map an online run leaves behind: points inserted the way NeuralPointCloud.add_neural_points does
    (src/neural_point.py:1557-1631) - pixels of the loop's views through the radius test against the cloud built so far
    (lk_add_points, radius_add 0.04, mapping.pixels_adding = 6 000 pixels per call as on Replica), three points per accepted ray at
    (0.98, 1.0, 1.02) x depth - until ~points_per_room points (~ 900 per m^2 of the 108 m^2 room: what the reference's 6 000 pixels per mapped frame
    converge to).  Larger maps are MORE ROOMS at the same density (copies of the room ROOM_PITCH apart with their own features):
    N grows with the explored area, the way a long ScanNet sequence grows its map - not by packing more points into one room.
    Returns (pos [N,3], geo [N,C], col [N,C], n_rooms) on eng.device."""
    from . import core, optim
    dev = eng.device
    n_rooms = max(1, int(round(n_points / points_per_room)))
    target = (n_points // n_rooms) // 3 * 3
    g = torch.Generator(device='cpu').manual_seed(seed)
    cap = target + 3 * pixels_per_call
    knn = core.KnnIndex(eng, capacity=cap)
    pos = torch.empty(cap, 3, device=dev)
    n, view, stall = 0, 0, 0
    r2 = float(torch.tensor(0.04 ** 2, dtype=torch.float32))          # pointcloud.radius_add, squared in fp32 as the ABI takes it
    while n < target and stall < 400:
        c2w = loop_pose(view % 200, 200, 'cpu')
        view += 7                                          # successive calls look from well separated poses of the loop
        i = torch.rand(pixels_per_call, generator=g) * (intr['W'] - 1)
        j = torch.rand(pixels_per_call, generator=g) * (intr['H'] - 1)
        ro, rd = pixel_rays(c2w, i, j, intr)
        d = room_depth(ro, rd)
        _, pts = optim.add_points(eng, knn if n > 0 else None, ro.to(dev), rd.to(dev), d.to(dev), r2, 0.98, 1.02, 3)
        k = min(int(pts.shape[0]), target - n)
        k -= k % 3
        stall = stall + 1 if k == 0 else 0
        if k:
            pos[n:n + k] = pts[:k]
            n += k
            knn.build(pos[:n])
    base = pos[:n].clone()
    knn.close()
    allpos = torch.cat([base + room_offset(r, dev) for r in range(n_rooms)], 0).contiguous()
    gd = torch.Generator(device=dev).manual_seed(seed + 1)
    geo = 0.1 * torch.randn(allpos.shape[0], c_dim, generator=gd, device=dev)
    col = 0.1 * torch.randn(allpos.shape[0], c_dim, generator=gd, device=dev)
    return allpos, geo, col, n_rooms


def default_weights(seed=1219, rel_pos=True, exposure=False):
    """Random-init decoder weights with the reference's shapes and init scheme
    (xavier_uniform(relu gain) trunk, default nn.Linear fc_c, B ~ N(0, scale^2); decoder.py:84-93,145-170,386-420)."""
    g = torch.Generator().manual_seed(seed)

    def xavier(out_f, in_f, gain):
        a = gain * math.sqrt(6.0 / (in_f + out_f))
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a

    def linear(out_f, in_f):
        a = 1.0 / math.sqrt(in_f)
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a, (torch.rand(out_f, generator=g) * 2 - 1) * a

    W = {}
    for pre, Hd, E in (('geo_decoder', 32, 93), ('color_decoder', 128, 40)):
        dims = [E, Hd, Hd, Hd + E, Hd]
        for i in range(5):
            W[f'{pre}.pts_linears.{i}.weight'] = xavier(Hd, dims[i], math.sqrt(2.0))
            W[f'{pre}.pts_linears.{i}.bias'] = torch.zeros(Hd)
            W[f'{pre}.fc_c.{i}.weight'], W[f'{pre}.fc_c.{i}.bias'] = linear(Hd, 32)
        n_out = 1 if pre == 'geo_decoder' else 3
        W[f'{pre}.output_linear.weight'] = xavier(n_out, Hd, math.sqrt(2.0) if n_out == 1 else 1.0)
        W[f'{pre}.output_linear.bias'] = torch.zeros(n_out)
    W['geo_decoder.embedder._B'] = torch.randn(3, 93, generator=g) * 25
    W['color_decoder.embedder._B'] = torch.randn(3, 20, generator=g) * 32
    W['color_decoder.embedder_rel_pos._B'] = torch.randn(3, 10, generator=g) * 32
    W['color_decoder.mlp_col_neighbor.linear1.weight'] = xavier(128, 52, 1.0)
    W['color_decoder.mlp_col_neighbor.linear1.bias'] = linear(128, 52)[1]
    W['color_decoder.mlp_col_neighbor.linear2.weight'] = xavier(32, 128, 1.0)
    W['color_decoder.mlp_col_neighbor.linear2.bias'] = linear(32, 128)[1]
    if exposure:        # MLP_exposure (decoder.py:326-342): 8 -> 128 -> 12, N(0, 0.01) weights, default nn.Linear biases
        W['color_decoder.mlp_exposure.linear1.weight'] = torch.randn(128, 8, generator=g) * 0.01
        W['color_decoder.mlp_exposure.linear1.bias'] = linear(128, 8)[1]
        W['color_decoder.mlp_exposure.linear2.weight'] = torch.randn(12, 128, generator=g) * 0.01
        W['color_decoder.mlp_exposure.linear2.bias'] = linear(12, 128)[1]
    return W
