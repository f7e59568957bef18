#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the .npz outputs are
committed, this script is committed, no reference file is copied anywhere.
Each fixture stores the inputs handed to the reference function and the outputs
(and autograd gradients) the reference produced on CPU (torch fp32).

    python tools/gen_golden.py            # rewrites tests/golden/*.npz

Fixtures:
  g1_composite       raw2outputs_nerf_color                      (common.py:382-422)
  g2_rays            get_rays_from_uv / get_rays                  (common.py:104-120,425-442)
  g3_pose            get_camera_from_tensor, get_tensor_from_camera (common.py:301-379)
  weights_<cfg>      NICER state_dict (+ colour embedder._B) under seed 1219
  g4_interp_<cfg>    MLP_*.get_feature_at_pos                     (decoder.py:180-231,431-492)
  g5_mlp_<cfg>       MLP_geometry.forward / MLP_color.forward     (decoder.py:233-288,494-546)
  g6_render_<cfg>_<mode>  Renderer.render_batch_ray fwd + autograd grads (Renderer.py:71-201)
                     with the Mapper / Tracker loss written as in Mapper.py:691-720 /
                     Tracker.py:169-191 (those are inline in the reference, so the
                     generator evaluates the same expressions on the reference's outputs)
  g8_adam            torch.optim.Adam trajectories with the stage lr schedule
  g9_add_points      NeuralPointCloud.add_neural_points (+ its find_neighbors_faiss radius rules)  (neural_point.py:1557-1631,1659-1708)
  g10_sample_near_pcl  NeuralPointCloud.sample_near_pcl           (neural_point.py:1734-1786)
  g11_filter_before_add  Mapper.filter_point_before_add           (Mapper.py:137-163)
  g12_keyframe_overlap   Mapper.keyframe_selection_overlap        (Mapper.py:219-282)
G9-G12 call the reference's methods UNBOUND on stand-in objects: the classes themselves need faiss-gpu, pydbow3 and the datasets to
construct, the methods are plain torch / numpy.  The stand-in index answers `search` exactly (FAISS-IVF itself stays unpinned).
"""
import os
import sys
import math

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import import_reference, cpu_get_device_patch, load_cfg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
SEED = 1219  # configs/point_slam.yaml:8

CFGS = {
    'replica': 'configs/Replica/room0.yaml',
    'tum': 'configs/TUM_RGBD/freiburg1_desk.yaml',
    'scannet': 'configs/ScanNet/scene0000.yaml',
}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)')


# ------------------------------------------------------------------ stand-in scene store
class ExactNPC:
    """Stand-in for NeuralPointCloud's query half: exact top-8 by squared L2.

    The reference's FAISS-GPU IVF index cannot be constructed here; this object
    offers the two methods the decoders call (find_neighbors_faiss, get_radius_query)
    with FAISS's return convention (D ascending fp32, I int64, count of D < r^2).
    """

    def __init__(self, pos, radius_query, k=8):
        self.pos = pos.float()
        self.rq = radius_query
        self.k = k

    def get_radius_query(self):
        return self.rq

    def find_neighbors_faiss(self, p, step='query', retrain=False, is_pts_grad=False, dynamic_radius=None):
        q = p.reshape(-1, 3).float()
        d = q[:, None, :] - self.pos[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        D, I = torch.topk(d2, self.k, dim=1, largest=False, sorted=True)
        if dynamic_radius is not None:
            nn = (D < dynamic_radius.reshape(-1, 1) ** 2).sum(-1).int()
        else:
            nn = (D < self.rq ** 2).sum(-1).int()
        return D, I, nn

    def device(self):
        return 'cpu'


class SlamNS:
    def __init__(self, cam):
        self.H, self.W = cam['H'], cam['W']
        self.fx, self.fy, self.cx, self.cy = cam['fx'], cam['fy'], cam['cx'], cam['cy']


def make_scene(g, R, N_extra, cam, radius, with_zero_depth=False):
    """Rays from a camera + a point cloud hugging the ray end points.

    Produces: rays with all samples supported, rays with none (invalid), samples with
    exactly one in-radius neighbour, and dense clusters (> 8 in radius)."""
    H, W, fx, fy, cx, cy = cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy']
    i = torch.randint(20, W - 20, (R,), generator=g).float()
    j = torch.randint(20, H - 20, (R,), generator=g).float()
    depth = 1.0 + 2.0 * torch.rand(R, generator=g)
    quat = torch.tensor([0.9, 0.1, -0.2, 0.05]) + 0.05 * torch.randn(4, generator=g)
    T = torch.tensor([0.3, -0.2, 0.5]) + 0.1 * torch.randn(3, generator=g)
    cam7 = torch.cat([quat, T])
    return i, j, depth, cam7


def cloud_for_rays(g, rays_o, rays_d, depth, radius):
    R = rays_o.shape[0]
    pts = []
    for r in range(R):
        mode = r % 8
        if mode == 7:
            continue                                   # unsupported ray -> invalid
        n_loc = [3, 3, 3, 1, 2, 6, 3][mode]
        ts = torch.tensor([0.98, 1.0, 1.02, 0.99, 1.01, 1.0])[:n_loc] if mode != 3 else torch.tensor([1.0])
        for t in ts:
            base = rays_o[r] + rays_d[r] * depth[r] * t
            m = 5 if mode == 5 else (1 if mode == 3 else 2)
            jit = (0.35 * radius) * torch.randn(m, 3, generator=g)
            pts.append(base[None, :] + jit)
    pts = torch.cat(pts, 0)
    # far-away clutter so the cloud is not only on rays
    clutter = torch.rand(600, 3, generator=g) * 6 - 3
    return torch.cat([pts, clutter], 0).float()


# ------------------------------------------------------------------ fixtures
def g1_composite(ref):
    g = torch.Generator().manual_seed(SEED)
    R, S = 64, 5
    raw = torch.randn(R, S, 4, generator=g)
    raw[..., 3] = raw[..., 3] * 40
    raw[5, :, 3] = -100.0
    raw[6, :3, 3] = -100.0
    d = 1 + 2 * torch.rand(R, 1, generator=g)
    z = d * torch.linspace(0.98, 1.02, S)[None, :]
    rays_d = torch.randn(R, 3, generator=g)
    depth, var, rgb, w = ref.common.raw2outputs_nerf_color(raw.clone(), z, rays_d, device='cpu', coef=0.1)
    save('g1_composite', raw=raw, z=z, coef=np.float32(0.1), depth=depth, var=var, rgb=rgb, w=w)


def g2_rays(ref, cfgs):
    out = {}
    g = torch.Generator().manual_seed(SEED + 2)
    for name, cfg in cfgs.items():
        cam = cfg['cam']
        i = torch.randint(0, cam['W'], (200,), generator=g).float()
        j = torch.randint(0, cam['H'], (200,), generator=g).float()
        c2w = torch.eye(4)
        A = torch.randn(3, 3, generator=g)
        Q, _ = torch.linalg.qr(A)
        c2w[:3, :3] = Q
        c2w[:3, 3] = torch.randn(3, generator=g)
        ro, rd = ref.common.get_rays_from_uv(i, j, c2w, cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy'], 'cpu')
        out[f'{name}_i'], out[f'{name}_j'], out[f'{name}_c2w'] = i, j, c2w
        out[f'{name}_intr'] = np.array([cam['fx'], cam['fy'], cam['cx'], cam['cy']], dtype=np.float64)
        out[f'{name}_rays_o'], out[f'{name}_rays_d'] = ro, rd
    # whole (small) image with crop
    c2w = out['tum_c2w']
    ro, rd = ref.common.get_rays(12, 16, 17.3, 16.5, 8.6, 5.3, c2w, 'cpu', crop_edge=2)
    out['img_rays_o'], out['img_rays_d'] = ro, rd
    out['img_params'] = np.array([12, 16, 17.3, 16.5, 8.6, 5.3, 2], dtype=np.float64)
    save('g2_rays', **out)


def g3_pose(ref):
    g = torch.Generator().manual_seed(SEED + 3)
    cams = torch.randn(16, 7, generator=g)
    cams[:, :4] *= torch.rand(16, 1, generator=g) * 2 + 0.2        # un-normalised quaternions
    with cpu_get_device_patch():
        c2w = ref.common.get_camera_from_tensor(cams)
        cam_g = cams[3].clone().requires_grad_(True)
        m = ref.common.get_camera_from_tensor(cam_g)
        Wt = torch.randn(3, 4, generator=g)
        (m * Wt).sum().backward()
    # inverse: 4x4 -> tensor
    back = []
    for n in range(16):
        M = torch.eye(4)
        M[:3] = c2w[n]
        back.append(ref.common.get_tensor_from_camera(M))
    save('g3_pose', cams=cams, c2w=c2w, back=torch.stack(back), grad_w=Wt, grad_cam3=cam_g.grad)


def build_model(ref, cfg):
    torch.manual_seed(SEED)
    model = ref.config.get_model(cfg)
    W = {k: v.detach().clone() for k, v in model.state_dict().items()}
    W['color_decoder.embedder._B'] = model.color_decoder.embedder._B.detach().clone()
    return model, W


def replay_noise(seed, P, C, two=True):
    """Re-draw what get_feature_at_pos draws (decoder.py:202-203,228-229 then 453-454,489-490)."""
    torch.manual_seed(seed)
    torch.zeros([P, C]).normal_(mean=0, std=0.01)
    n_geo = torch.zeros([C]).normal_(mean=0, std=0.01)
    if not two:
        return n_geo, None
    torch.zeros([P, C]).normal_(mean=0, std=0.01)
    n_col = torch.zeros([C]).normal_(mean=0, std=0.01)
    return n_geo, n_col


def g456(ref, name, cfg):
    """Interp (g4), decoders (g5) and end-to-end render + grads (g6) for one config."""
    model, W = build_model(ref, cfg)
    save(f'weights_{name}', **W)
    cam = cfg['cam']
    C = cfg['model']['c_dim']
    S = cfg['rendering']['N_surface']
    dyn = cfg['use_dynamic_radius']
    rq = cfg['pointcloud']['radius_query']
    g = torch.Generator().manual_seed(SEED + 10 + len(name))
    R = 96
    i, j, depth, cam7 = make_scene(g, R, 0, cam, rq)
    with cpu_get_device_patch():
        c2w0 = ref.common.get_camera_from_tensor(cam7)
    ro0, rd0 = ref.common.get_rays_from_uv(i, j, c2w0, cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy'], 'cpu')
    pos = cloud_for_rays(g, ro0, rd0, depth, rq)
    N = pos.shape[0]
    geo = (0.1 * torch.randn(N, C, generator=g))
    col = (0.1 * torch.randn(N, C, generator=g))
    gt_color = torch.rand(R, 3, generator=g)
    # dynamic radius is float64 in the reference (interp1d output, Tracker.py:255-258)
    r_query = (0.04 + 0.12 * torch.rand(R, generator=g, dtype=torch.float64)) if dyn else None
    exposure = cfg['model']['encode_exposure']
    exp_feat = (0.3 * torch.randn(cfg['model']['exposure_dim'], generator=g)) if exposure else None
    if exposure:  # default init is N(0, 0.01): make the affine non-trivial
        with torch.no_grad():
            model.color_decoder.mlp_exposure.linear2.bias.copy_(torch.tensor(
                [1., 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]) + 0.05 * torch.randn(12, generator=g))
        W = {k: v.detach().clone() for k, v in model.state_dict().items()}
        W['color_decoder.embedder._B'] = model.color_decoder.embedder._B.detach().clone()
        save(f'weights_{name}', **W)

    npc = ExactNPC(pos, rq)
    renderer = ref.renderer.Renderer(cfg, None, SlamNS(cam))
    renderer.sigmoid_coefficient = cfg['rendering']['sigmoid_coef_mapper']

    # ---------------- g4 / g5 on the sample points of the rays
    z = depth.reshape(-1, 1) * (cfg['rendering']['near_end_surface'] * (1 - torch.linspace(0, 1, S)) +
                                cfg['rendering']['far_end_surface'] * torch.linspace(0, 1, S))[None, :]
    p = (ro0[:, None, :] + rd0[:, None, :] * z[..., None]).reshape(-1, 3)
    P = p.shape[0]
    r_pts = r_query.reshape(-1, 1).repeat_interleave(S, dim=0) if dyn else None
    D, I, nn = npc.find_neighbors_faiss(p, dynamic_radius=r_pts)
    g4 = dict(p=p, pos=pos, geo=geo, col=col, D=D, I=I, nn=nn)
    if dyn:
        g4['r_pts'] = r_pts
    for trk in (False, True):
        torch.manual_seed(77)
        cg, _, hasg = model.geo_decoder.get_feature_at_pos(npc, p, geo, is_tracker=trk, cloud_pos=pos, dynamic_r_query=r_pts)
        cc, _, hasc = model.color_decoder.get_feature_at_pos(npc, p, col, is_tracker=trk, cloud_pos=pos, dynamic_r_query=r_pts)
        ng, nc = replay_noise(77, P, C)
        g4[f'c_geo_trk{int(trk)}'], g4[f'c_col_trk{int(trk)}'] = cg, cc
        g4[f'has_trk{int(trk)}'] = hasg
        g4['noise_geo'], g4['noise_col'] = ng, nc
    save(f'g4_interp_{name}', **g4)

    torch.manual_seed(78)
    occ, vray, has = model.geo_decoder(p, npc, geo, col, pts_num=S, is_tracker=False, cloud_pos=pos, dynamic_r_query=r_pts)
    rgb = model.color_decoder(p, npc, geo, col, is_tracker=False, cloud_pos=pos, dynamic_r_query=r_pts, exposure_feat=exp_feat)
    ng, nc = replay_noise(78, P, C)
    g5 = dict(p=p, occ=occ, valid_ray=vray, has=has, rgb=rgb, noise_geo=ng, noise_col=nc,
              c_geo=g4['c_geo_trk0'], c_col=None)
    # decoder outputs as a pure function of (p, c): recompute c under the same seed
    torch.manual_seed(78)
    cg, _, _ = model.geo_decoder.get_feature_at_pos(npc, p, geo, False, pos, dynamic_r_query=r_pts)
    cc, _, _ = model.color_decoder.get_feature_at_pos(npc, p, col, False, pos, dynamic_r_query=r_pts)
    g5['c_geo'], g5['c_col'] = cg, cc
    if exposure:
        g5['exposure_feat'] = exp_feat
        g5['affine'] = model.color_decoder.mlp_exposure(exp_feat)
        rgb_raw = model.color_decoder(p, npc, geo, col, is_tracker=False, cloud_pos=pos, dynamic_r_query=r_pts, exposure_feat=None)
        g5['rgb_logits'] = rgb_raw
    save(f'g5_mlp_{name}', **g5)

    # ---------------- g6: end-to-end, mapper mode (grads to feats + decoder params)
    for stage in ('geometry', 'color'):
        for prm in model.parameters():
            prm.grad = None
        geo_l = geo.clone().requires_grad_(True)
        col_l = col.clone().requires_grad_(True)
        torch.manual_seed(90)
        if stage == 'geometry':
            # NICER.forward(stage='geometry') builds a 'cuda:-1' device string on CPU
            # (decoder.py:591,597); assemble raw exactly as decoder.py:594-600 does.
            class GeoOnly(torch.nn.Module):
                def __init__(s, m):
                    super().__init__()
                    s.m = m

                def forward(s, p_, npc_, stage_, gf, cf, pts_num, is_tracker, cloud_pos, pts_views_d, r_q, exp):
                    occ_, rm, pm = s.m.geo_decoder(p_, npc_, gf, cf, pts_num=pts_num, is_tracker=is_tracker,
                                                   cloud_pos=cloud_pos, dynamic_r_query=r_q)
                    raw = torch.zeros(occ_.shape[0], 4, dtype=torch.float)
                    raw[..., -1] = occ_
                    return raw, rm, pm
            dec = GeoOnly(model)
        else:
            dec = model
        d, u, c, vm = renderer.render_batch_ray(npc, dec, rd0, ro0, 'cpu', stage, gt_depth=depth,
                                                npc_geo_feats=geo_l, npc_col_feats=col_l, is_tracker=False,
                                                cloud_pos=pos, dynamic_r_query=r_query, exposure_feat=None)
        ng, nc = replay_noise(90, P, C, two=(stage == 'color'))
        # Mapper.py:691-720 (non-exposure branch) evaluated on the reference's outputs
        dm = (depth > 0) & vm & (~torch.isnan(d))
        geo_loss = torch.abs(depth[dm] - d[dm]).sum()
        loss = geo_loss.clone()
        color_loss = torch.zeros(())
        if stage == 'color' and not exposure:
            color_loss = torch.abs(gt_color[dm] - c[dm]).sum()
            loss = loss + cfg['mapping']['w_color_loss'] * color_loss
        elif stage == 'color' and exposure:
            # mapper exposure path (Mapper.py:697-715): one frame -> one affine on pre-sigmoid colour
            aff = model.color_decoder.mlp_exposure(exp_feat)
            c2 = torch.sigmoid(torch.matmul(c, aff[:9].reshape(3, 3)) + aff[-3:])
            color_loss = torch.abs(gt_color[dm] - c2[dm]).sum()
            loss = loss + cfg['mapping']['w_color_loss'] * color_loss
        loss.backward()
        out = dict(rays_o=ro0, rays_d=rd0, gt_depth=depth, gt_color=gt_color, pos=pos, geo=geo, col=col,
                   depth=d, var=u, color=c, valid_ray=vm, loss=loss, geo_loss=geo_loss, color_loss=color_loss,
                   noise_geo=ng, grad_geo=geo_l.grad,
                   w_color=np.float32(cfg['mapping']['w_color_loss']))
        if nc is not None:
            out['noise_col'] = nc
        if col_l.grad is not None:
            out['grad_col'] = col_l.grad
        if dyn:
            out['r_query'] = r_query
        if exposure:
            out['exposure_feat'] = exp_feat
        for k_, prm in model.named_parameters():
            if prm.grad is not None:
                out['gradW.' + k_] = prm.grad
        save(f'g6_render_{name}_map_{stage}', **out)

    # ---------------- g6: tracker mode (grad to the 7-vector pose through rays)
    for prm in model.parameters():
        prm.grad = None
    renderer.sigmoid_coefficient = cfg['rendering']['sigmoid_coef_tracker']
    cam_l = cam7.clone().requires_grad_(True)
    exp_l = exp_feat.clone().requires_grad_(True) if exposure else None
    with cpu_get_device_patch():
        c2w = ref.common.get_camera_from_tensor(cam_l)
    ro, rd = ref.common.get_rays_from_uv(i, j, c2w, cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy'], 'cpu')
    ro = ro.clone()
    ro.retain_grad()
    rd.retain_grad()
    torch.manual_seed(91)
    d, u, c, vm = renderer.render_batch_ray(npc, model, rd, ro, 'cpu', 'color', gt_depth=depth,
                                            npc_geo_feats=geo, npc_col_feats=col, is_tracker=True,
                                            cloud_pos=pos, dynamic_r_query=r_query, exposure_feat=exp_l)
    ng, nc = replay_noise(91, P, C)
    # Tracker.py:169-191 (handle_dynamic, use_color_in_tracking) on the reference's outputs
    unc = u.detach()
    nan_mask = (~torch.isnan(d)) & (~torch.isnan(unc))
    tmp = torch.abs(depth - d) / torch.sqrt(unc + 1e-10)
    mask = (tmp < 10 * tmp.mean()) & (depth > 0) & nan_mask
    geo_loss = torch.clamp(tmp, min=0.0, max=1e3)[mask].sum()
    color_loss = torch.abs(gt_color - c)[mask].sum()
    loss = geo_loss + cfg['tracking']['w_color_loss'] * color_loss
    loss.backward()
    out = dict(cam=cam7, i=i, j=j, intr=np.array([cam['fx'], cam['fy'], cam['cx'], cam['cy']], dtype=np.float64),
               rays_o=ro, rays_d=rd, gt_depth=depth, gt_color=gt_color, pos=pos, geo=geo, col=col,
               depth=d, var=u, color=c, valid_ray=vm, mask=mask, loss=loss, geo_loss=geo_loss, color_loss=color_loss,
               noise_geo=ng, noise_col=nc, grad_cam=cam_l.grad, grad_rays_o=ro.grad, grad_rays_d=rd.grad,
               w_color=np.float32(cfg['tracking']['w_color_loss']))
    if dyn:
        out['r_query'] = r_query
    if exposure:
        out['exposure_feat'] = exp_feat
        out['grad_exposure_feat'] = exp_l.grad
        for k_, prm in model.color_decoder.mlp_exposure.named_parameters():
            out['gradW.color_decoder.mlp_exposure.' + k_] = prm.grad
    save(f'g6_render_{name}_track', **out)

    # ---------------- g6: full-image style batch with zero-depth rays (render_img path, no grad)
    with torch.no_grad():
        depth_z = depth.clone()
        depth_z[::5] = 0.0
        torch.manual_seed(92)
        renderer.sigmoid_coefficient = cfg['rendering']['sigmoid_coef_mapper']
        d, u, c, vm = renderer.render_batch_ray(npc, model, rd0, ro0, 'cpu', 'color', gt_depth=depth_z,
                                                npc_geo_feats=geo, npc_col_feats=col, is_tracker=False,
                                                cloud_pos=pos, dynamic_r_query=r_query, exposure_feat=exp_feat)
        ng, nc = replay_noise(92, P, C)
    out = dict(rays_o=ro0, rays_d=rd0, gt_depth=depth_z, pos=pos, geo=geo, col=col, depth=d, var=u, color=c,
               valid_ray=vm, noise_geo=ng, noise_col=nc)
    if dyn:
        out['r_query'] = r_query
    if exposure:
        out['exposure_feat'] = exp_feat
    save(f'g6_render_{name}_img', **out)


def g8_adam():
    """torch.optim.Adam trajectories: 3 groups with the geometry->colour lr switch
    (Mapper.py:562-607) and a parameter that has no grad during the first stage."""
    g = torch.Generator().manual_seed(SEED + 8)
    dec = torch.randn(50, generator=g).requires_grad_(True)
    geo = torch.randn(40, 32, generator=g).requires_grad_(True)
    col = torch.randn(40, 32, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([{'params': [dec], 'lr': 0}, {'params': [geo], 'lr': 0}, {'params': [col], 'lr': 0}])
    p0 = [dec.detach().clone(), geo.detach().clone(), col.detach().clone()]
    grads, traj = [], []
    for it in range(20):
        stage_geo = it <= 7
        opt.param_groups[0]['lr'] = 0.001 if stage_geo else 0.005
        opt.param_groups[1]['lr'] = 0.03 if stage_geo else 0.005
        opt.param_groups[2]['lr'] = 0.0 if stage_geo else 0.005
        opt.zero_grad()
        gd = torch.randn(50, generator=g)
        gg = torch.randn(40, 32, generator=g) * (torch.rand(40, 1, generator=g) > 0.3)   # untouched rows
        gc = torch.randn(40, 32, generator=g)
        dec.grad, geo.grad = gd.clone(), gg.clone()
        col.grad = None if stage_geo else gc.clone()          # colour rows have no grad in stage geometry
        opt.step()
        grads.append((gd, gg, gc))
        traj.append((dec.detach().clone(), geo.detach().clone(), col.detach().clone()))
    save('g8_adam', dec0=p0[0], geo0=p0[1], col0=p0[2],
         gd=torch.stack([x[0] for x in grads]), gg=torch.stack([x[1] for x in grads]), gc=torch.stack([x[2] for x in grads]),
         dec=torch.stack([x[0] for x in traj]), geo=torch.stack([x[1] for x in traj]), col=torch.stack([x[2] for x in traj]),
         n_geo_stage=np.int32(8))


# ------------------------------------------------------------------ G9-G12: map maintenance, reference methods on stand-ins
class ExactIndex:
    """Stand-in for the FAISS IVF index of NeuralPointCloud: exact squared-L2 top-k over the stored points, FAISS's
    conventions (ascending D, missing slots D = FLT_MAX / I = -1, is_trained)."""

    def __init__(self, pos=None):
        self.pos = torch.zeros(0, 3) if pos is None else pos.float().clone()
        self.is_trained = self.pos.shape[0] > 0

    def train(self, x):
        self.is_trained = True

    def add(self, x):
        self.pos = torch.cat([self.pos, torch.as_tensor(x).float().reshape(-1, 3)], 0)

    def search(self, q, k):
        q = q.reshape(-1, 3).float()
        n = self.pos.shape[0]
        d = q[:, None, :] - self.pos[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        kk = min(k, n)
        D, I = torch.topk(d2, kk, dim=1, largest=False, sorted=True)
        if kk < k:
            D = torch.cat([D, torch.full((q.shape[0], k - kk), float(np.finfo(np.float32).max))], 1)
            I = torch.cat([I, torch.full((q.shape[0], k - kk), -1, dtype=I.dtype)], 1)
        return D, I


def npc_standin(NPC, cfg, pos):
    """An object with the attributes NeuralPointCloud.__init__ reads from the config (neural_point.py:30-90) that the three
    methods touch; find_neighbors_faiss is the REFERENCE's own method bound to it."""
    import types
    pc = cfg['pointcloud']
    o = types.SimpleNamespace()
    o.device = 'cpu'
    o.c_dim, o.nn_num = cfg['model']['c_dim'], pc['nn_num']
    o.radius_add, o.radius_min, o.radius_query, o.radius_mesh = pc['radius_add'], pc['radius_min'], pc['radius_query'], pc.get('radius_mesh', 0.08)
    o.N_add, o.near_end_surface, o.far_end_surface = pc['N_add'], pc['near_end_surface'], pc['far_end_surface']
    o.fix_interval_when_add_along_ray = pc.get('fix_interval_when_add_along_ray', False)
    o.segment_strategy = 'rot_trans'
    o._input_pos, o._input_rgb, o._pts_num = [], [], 0
    o.index = ExactIndex(pos)
    o.captured = {}
    o.check_index = lambda method=None, idx=None, cur_c2w=None: False
    o.update_fragments = lambda **kw: o.captured.update(kw)
    o.get_cloud_pos = lambda end=False: o.index.pos.tolist()
    o.find_neighbors_faiss = types.MethodType(NPC.find_neighbors_faiss, o)
    return o


def g9_add_points(cfg):
    import src.neural_point as npm
    NPC = npm.NeuralPointCloud
    g = torch.Generator().manual_seed(SEED + 9)
    n = 400
    ro = torch.tensor([0.2, -0.1, 0.3]).repeat(n, 1)
    rd = torch.cat([torch.rand(n, 2, generator=g) * 1.2 - 0.6, -torch.ones(n, 1)], 1)
    gd = 1.0 + 2.0 * torch.rand(n, generator=g)
    gc = torch.rand(n, 3, generator=g)
    # existing cloud: near the surface points of about half of the rays, at distances around the three radii
    surf = ro + rd * gd[:, None]
    pick = torch.randperm(n, generator=g)[: n // 2]
    dirs = torch.nn.functional.normalize(torch.randn(pick.numel(), 3, generator=g), dim=1)
    dist = torch.rand(pick.numel(), 1, generator=g) * 0.1                      # 0 .. 0.1 m: straddles radius_min 0.02, radius_add 0.04, 0.08
    cloud = torch.cat([surf[pick] + dirs * dist, torch.rand(300, 3, generator=g) * 6 - 3], 0)
    out = dict(rays_o=ro, rays_d=rd, gt_color=gc, cloud=cloud)
    gd_holes = gd.clone(); gd_holes[::17] = 0.0                                # rays without a depth reading are dropped first
    dyn = (0.02 + 0.06 * torch.rand(n, generator=g)).double()                  # per-ray dynamic radius_add (float64, Mapper.py:854-872)
    cases = {'static': dict(depth=gd_holes, kw={}), 'grad': dict(depth=gd_holes, kw=dict(is_pts_grad=True)),
             'dynamic': dict(depth=gd, kw=dict(dynamic_radius=dyn)), 'empty': dict(depth=gd_holes, kw={}, empty=True)}
    for name, c in cases.items():
        o = npc_standin(NPC, cfg, None if c.get('empty') else cloud)
        torch.manual_seed(3)
        ret = NPC.add_neural_points(o, ro.clone(), rd.clone(), c['depth'].clone(), gc.clone(), **c['kw'])
        out[f'{name}_depth'] = c['depth']
        out[f'{name}_count'] = np.int64(int(ret))
        out[f'{name}_new_points'] = o.captured['npc']                          # what update_fragments receives as the new neural points
        out[f'{name}_input_pos'] = torch.tensor(o._input_pos, dtype=torch.float32).reshape(-1, 3)
        out[f'{name}_input_rgb'] = torch.tensor(o._input_rgb, dtype=torch.float32).reshape(-1, 3)
        out[f'{name}_index_size'] = np.int64(o.index.pos.shape[0])
    out['dynamic_radius'] = dyn
    out['radius_add'], out['radius_min'] = np.float64(cfg['pointcloud']['radius_add']), np.float64(cfg['pointcloud']['radius_min'])
    out['near_surface'], out['far_surface'] = np.float64(cfg['pointcloud']['near_end_surface']), np.float64(cfg['pointcloud']['far_end_surface'])
    save('g9_add_points', **out)


def g10_sample_near_pcl(cfg):
    import src.neural_point as npm
    NPC = npm.NeuralPointCloud
    g = torch.Generator().manual_seed(SEED + 10)
    wall = torch.cat([torch.rand(2500, 2, generator=g) * 4 - 2, torch.full((2500, 1), 2.0)], 1)       # a wall at z = 2
    wall2 = torch.cat([torch.rand(900, 2, generator=g) * 4 - 2, torch.full((900, 1), 3.1)], 1)        # a sparser one behind it
    cloud = torch.cat([wall, wall2], 0) + 0.01 * torch.randn(3400, 3, generator=g)
    n = 60
    ro = torch.zeros(n, 3)
    rd = torch.cat([torch.rand(n, 2, generator=g) * 0.8 - 0.4, torch.ones(n, 1)], 1)
    rd[:6, 2] = -1.0                                                            # looking away from the cloud: invalid rays
    o = npc_standin(NPC, cfg, cloud)
    near, far, num = 0.3, torch.tensor(4.0), 5
    z, invalid = NPC.sample_near_pcl(o, ro, rd, near, far, num)
    save('g10_sample_near_pcl', rays_o=ro, rays_d=rd, cloud=cloud, near=np.float64(near), far=np.float64(4.0), num=np.int64(num),
         radius_query=np.float64(cfg['pointcloud']['radius_query']), z=z, invalid=invalid)


def mapper_standin(cam):
    import types
    o = types.SimpleNamespace()
    o.device = 'cpu'
    o.H, o.W, o.fx, o.fy, o.cx, o.cy = cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy']
    return o


def _pose(g, yaw, t):
    """c2w looking down -z, rotated by `yaw` about y, at position t (the dataset convention after the readers' flip)."""
    c, s_ = math.cos(yaw), math.sin(yaw)
    M = torch.eye(4)
    M[:3, :3] = torch.tensor([[c, 0., s_], [0., 1., 0.], [-s_, 0., c]])
    M[:3, 3] = torch.tensor(t)
    return M


def g11_filter_before_add(ref, cfg):
    import src.Mapper as M
    cam = cfg['cam']
    g = torch.Generator().manual_seed(SEED + 11)
    o = mapper_standin(cam)
    cur, prev = _pose(g, 0.25, [0.1, 0.0, 0.2]), _pose(g, -0.15, [0.0, 0.05, 0.0])
    n = 500
    i = torch.randint(0, cam['W'], (n,), generator=g).float()
    j = torch.randint(0, cam['H'], (n,), generator=g).float()
    ro, rd = ref.common.get_rays_from_uv(i, j, cur, cam['H'], cam['W'], cam['fx'], cam['fy'], cam['cx'], cam['cy'], 'cpu')
    gd = 0.5 + 3.0 * torch.rand(n, generator=g)
    mask = M.Mapper.filter_point_before_add(o, ro, rd, gd, prev)
    save('g11_filter_before_add', rays_o=ro, rays_d=rd, gt_depth=gd, prev_c2w=prev, cur_c2w=cur, outside=mask,
         intr=np.array([cam['fx'], cam['fy'], cam['cx'], cam['cy']], np.float64), HW=np.array([cam['H'], cam['W']], np.int64))


def g12_keyframe_overlap(ref, cfg):
    import builtins
    import src.Mapper as M
    cam = cfg['cam']
    H, W = cam['H'], cam['W']
    g = torch.Generator().manual_seed(SEED + 12)
    o = mapper_standin(cam)
    cur = _pose(g, 0.1, [0.0, 0.0, 0.0])
    yaws = [0.0, 0.3, 0.8, 1.6, 3.0, -0.5, 0.12]
    kfs = [{'est_c2w': _pose(g, y, [0.2 * k - 0.5, 0.03 * k, 0.1 * k])} for k, y in enumerate(yaws)]
    depth = 1.5 + torch.rand(H, W, generator=g)
    depth[::9, ::7] = 0.0
    color = torch.rand(H, W, 3, generator=g)
    rec = {}
    orig_get_samples = M.get_samples

    def rec_samples(*a, **k):                       # the reference's own draw, recorded (its pixels come from the global RNG)
        r = orig_get_samples(*a, **k)
        rec['rays_o'], rec['rays_d'], rec['gt_depth'] = r[0].clone(), r[1].clone(), r[2].clone()
        return r

    def rec_sorted(lst, **k):                       # percent_inside of every keyframe, as handed to sorted() (Mapper.py:274-275)
        rec['percent'] = [float(d['percent_inside']) for d in lst]
        return builtins.sorted(lst, **k)
    M.get_samples, M.sorted = rec_samples, rec_sorted
    try:
        torch.manual_seed(SEED)
        np.random.seed(SEED)
        sel = M.Mapper.keyframe_selection_overlap(o, color, depth, cur, kfs, k=len(kfs))
    finally:
        M.get_samples = orig_get_samples
        del M.sorted
    save('g12_keyframe_overlap', rays_o=rec['rays_o'], rays_d=rec['rays_d'], gt_depth=rec['gt_depth'],
         est_c2ws=torch.stack([kf['est_c2w'] for kf in kfs]), percent_inside=np.asarray(rec['percent'], np.float64),
         selected=np.asarray(sorted(int(x) for x in sel), np.int64), N_samples=np.int64(8),
         intr=np.array([cam['fx'], cam['fy'], cam['cx'], cam['cy']], np.float64), HW=np.array([H, W], np.int64))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)            # deterministic reductions
    ref = import_reference()
    cfgs = {k: load_cfg(ref, v) for k, v in CFGS.items()}
    print('g1'); g1_composite(ref)
    print('g2'); g2_rays(ref, cfgs)
    print('g3'); g3_pose(ref)
    for name, cfg in cfgs.items():
        print('g4-6', name)
        g456(ref, name, cfg)
    print('g8'); g8_adam()
    maintenance(ref, cfgs)


def maintenance(ref, cfgs):
    print('g9-12')
    g9_add_points(cfgs['replica'])
    g10_sample_near_pcl(cfgs['replica'])
    g11_filter_before_add(ref, cfgs['tum'])
    g12_keyframe_overlap(ref, cfgs['tum'])


if __name__ == '__main__':
    if '--maintenance-only' in sys.argv:            # G9-G12 alone (the other fixtures are unchanged)
        torch.set_num_threads(1)
        _ref = import_reference()
        maintenance(_ref, {k: load_cfg(_ref, v) for k, v in CFGS.items()})
    else:
        main()
