"""The two per-frame optimisation loops of the hot path, as launch sequences over the C ABI.

  MapOptimizer    inner loop of Mapper.optimize_map   (src/Mapper.py:562-735)
  TrackOptimizer  inner loop of Tracker.run / optimize_cam_in_batch (src/Tracker.py:102-197, 313-401)

Everything inside an iteration stays on the device and has static shapes, so there is no host
synchronisation in the loop:
  * rays the reference drops by boolean indexing (depth <= 0, outside the inside-mask) are kept in the
    batch with gt_depth := 0 — "absent" for the losses (lk_inside_mask / lk_loss_*), identical sums;
  * the frustum-selected feature rows are optimised IN PLACE in the full tables through a row index
    (lk_adam_step row_index) instead of the reference's gather / index_put round trip
    (Mapper.py:578-586,727-735), with torch.optim.Adam's dense semantics on exactly those rows;
  * the tracker's best-loss candidate (Tracker.py:375-377) is chosen at the end from a device-side log.
"""
import torch

from . import _ffi, core, optim

GEO_DECODER_PARAMS = ('geo_decoder.embedder._B',)      # fix_geo_decoder: True (Mapper.py:537-541)
COLOR_DECODER_PARAMS = tuple(
    [f'color_decoder.pts_linears.{i}.{w}' for i in range(5) for w in ('weight', 'bias')] +
    [f'color_decoder.fc_c.{i}.{w}' for i in range(5) for w in ('weight', 'bias')] +
    ['color_decoder.output_linear.weight', 'color_decoder.output_linear.bias',
     'color_decoder.embedder_rel_pos._B',
     'color_decoder.mlp_col_neighbor.linear1.weight', 'color_decoder.mlp_col_neighbor.linear1.bias',
     'color_decoder.mlp_col_neighbor.linear2.weight', 'color_decoder.mlp_col_neighbor.linear2.bias'])


class RayBatch:
    """Device buffers of one ray batch (static size R)."""

    def __init__(self, eng, R, dynamic_radius=False):
        e = eng
        self.R = R
        self.rays_o, self.rays_d = e.empty(R, 3), e.empty(R, 3)
        self.gt_depth, self.gt_color = e.empty(R), e.empty(R, 3)
        self.pix_i, self.pix_j = e.empty(R), e.empty(R)
        self.r2_ray = e.empty(R) if dynamic_radius else None
        self.d_depth, self.d_color = e.empty(R), e.empty(R, 3)
        self.thr = e.empty(1)
        self.scratch_u32 = e.empty(R, dtype=torch.int32)
        self.loss_scratch = e.empty(R + 8)

    def as_out(self):
        d = dict(rays_o=self.rays_o, rays_d=self.rays_d, gt_depth=self.gt_depth, gt_color=self.gt_color,
                 pix_i=self.pix_i, pix_j=self.pix_j)
        if self.r2_ray is not None:
            d['r2_ray'] = self.r2_ray
        return d


class MapOptimizer:
    """One optimize_map call: Adam over {decoder params, selected geo rows, selected colour rows}."""

    def __init__(self, eng, cfg, dec, knn, pos, geo_feats, col_feats, row_index, R, lrs, w_color=0.1,
                 dynamic_radius=False, fix_color_decoder=False, dist=None, exposure=None):
        """row_index: int32 [n_f] rows being optimised (frustum selection, Mapper.py:498-512) or None = all rows.
        lrs: dict stage -> (decoders_lr, geometry_lr, color_lr)  (configs mapping.stage.*)."""
        self.eng, self.cfg, self.dec, self.knn = eng, cfg, dec, knn
        self.pos, self.geo, self.col = pos, geo_feats, col_feats
        self.rows = row_index
        self.lrs, self.w_color = lrs, w_color
        self.R = R
        self.batch = RayBatch(eng, R, dynamic_radius)
        self.st = core.RenderState(eng, R, cfg.S, need_act=True)
        N = geo_feats.shape[0]
        self.gs = core.GradState(eng, N, R, dec.n, feats=True, weights=True)
        self.adam = optim.Adam(eng)            # fresh optimiser per optimize_map call (Mapper.py:570)
        # torch.optim.Adam skips parameters whose .grad is None and keeps a step count PER parameter: the colour
        # decoder receives no gradient in stage 'geometry', so its ranges only enter the colour-stage steps
        cnames = [n for n in COLOR_DECODER_PARAMS if n in dec.layout] if not fix_color_decoder else \
            ['color_decoder.embedder_rel_pos._B']
        if not cfg.rel_pos:
            cnames = [n for n in cnames if 'mlp_col_neighbor' not in n and 'embedder_rel_pos' not in n]
        self.geo_dec_ranges = dec.param_ranges(list(GEO_DECODER_PARAMS))
        self.col_dec_ranges = dec.param_ranges(cnames)
        self.loss_log = None
        self.dist = dist
        self.it = 0
        # exposure = (mlp_exposure torch module, [exposure_feat tensor per frame of the window]) for model.encode_exposure
        # (ScanNet): per-keyframe colour affine applied to the RENDERED colour logits (Mapper.py:697-715)
        self.exposure = exposure
        self.exp_opt = None

    def iterate(self, stage, frames, rnd, frame_id, window, intr, H, W, log_row=None):
        """One joint iteration (Mapper.py:576-735).
        frames: (depth_stack [F,H,W], color_stack [F,H,W,3], c2w_stack [F,4,4], r2_map_stack or None);
        rnd int32 [R] window-pixel draws, frame_id int32 [R]."""
        eng, b, st, gs = self.eng, self.batch, self.st, self.gs
        depth_stack, color_stack, c2w_stack, r2_stack = frames
        optim.gather_rays(eng, depth_stack, color_stack, c2w_stack, frame_id, rnd, H, W, window, intr, b.as_out(), r2_stack)
        optim.inside_mask(eng, b.gt_depth, None, b.thr, b.scratch_u32, depth_filtered=b.gt_depth)
        core.render_forward(eng, self.cfg, st, b.rays_o, b.rays_d, b.gt_depth, self.knn, self.pos, self.geo, self.col,
                            self.dec, stage, r2_ray=b.r2_ray, save_act=True, extra_flags=_ffi.FLAG_ZERO_ABSENT,
                            color_logits=self.exposure is not None)
        out4 = log_row if log_row is not None else self._out4()
        if self.exposure is not None and stage == 'color':
            self._exposure_loss(st, b, frame_id, out4)
        else:
            optim.loss_mapper(eng, st, b.gt_depth, b.gt_color, self.w_color, stage == 'color', b.d_depth, b.d_color, out4)
        core.render_backward(eng, st, gs, b.d_depth, b.d_color)
        if self.dist is not None:
            self.dist.all_reduce_grads(self, stage)
        dlr, glr, clr = self.lrs[stage]
        segs = [(('gdec', k), self.dec.blob[o:o + n], gs.g_weights[o:o + n], dlr) for k, (o, n) in enumerate(self.geo_dec_ranges)]
        if stage == 'color':
            segs += [(('cdec', k), self.dec.blob[o:o + n], gs.g_weights[o:o + n], dlr) for k, (o, n) in enumerate(self.col_dec_ranges)]
        if self.rows is not None:
            segs.append(('geo', self.geo, gs.g_geo, glr, self.rows))
            if stage == 'color':          # colour rows have no gradient in stage 'geometry' -> skipped by torch Adam
                segs.append(('col', self.col, gs.g_col, clr, self.rows))
        else:
            segs.append(('geo', self.geo.view(-1), gs.g_geo.view(-1), glr))
            if stage == 'color':
                segs.append(('col', self.col.view(-1), gs.g_col.view(-1), clr))
        self.adam.step(segs, zero_grad=True)
        if stage == 'color':
            # the geometry stage only moves the embedding matrices (read from the plain blob); the MFMA fragments are
            # copies of the colour-decoder matrices, which only change in the colour stage
            self.dec.repack()
        self.it += 1
        return out4

    def _exposure_loss(self, st, b, frame_id, out4):
        """Colour stage with exposure encoding (Mapper.py:691-720): the renderer returned colour LOGITS; the rays of
        keyframe f get sigmoid(logits @ rot_f + trans_f), (rot_f | trans_f) = mlp_exposure(exposure_feat_f).  This per-ray
        epilogue and the tiny exposure MLP run in torch autograd on [R,3] tensors; what leaves it are d depth / d colour
        logits for lk_render_bwd and an Adam step (lr 1e-3, Mapper.py:600-607) on the exposure features and the MLP."""
        mlp, feats = self.exposure
        if self.exp_opt is None:
            self.exp_opt = torch.optim.Adam([{'params': feats, 'lr': 0.001}, {'params': list(mlp.parameters()), 'lr': 0.001}])
        depth = st.depth.detach()
        color = st.color.detach().clone().requires_grad_(True)
        aff = torch.stack([mlp(f) for f in feats])                               # [F,12]
        fid = frame_id.long() if frame_id is not None else torch.zeros(color.shape[0], dtype=torch.long, device=color.device)
        rot, trans = aff[:, :9].reshape(-1, 3, 3)[fid], aff[:, 9:][fid]
        col = torch.sigmoid(torch.einsum('rc,rcd->rd', color, rot) + trans)
        m = (b.gt_depth > 0) & st.valid_ray.bool() & (~torch.isnan(depth))
        geo = torch.abs(b.gt_depth - depth)[m].sum()
        closs = torch.abs(b.gt_color - col)[m].sum()
        loss = geo + self.w_color * closs
        self.exp_opt.zero_grad()
        (self.w_color * closs).backward()
        self.exp_opt.step()
        b.d_color.copy_(color.grad)
        b.d_depth.copy_(torch.where(m, torch.sign(depth - b.gt_depth), torch.zeros_like(depth)))
        out4.copy_(torch.stack([loss.detach(), geo.detach(), closs.detach(), m.sum().float()]))

    def _out4(self):
        if self.loss_log is None:
            self.loss_log = self.eng.zeros(4)
        return self.loss_log

    def new_frame(self, row_index, row_mask=None):
        """Start of an optimize_map call: the frustum rows of the new frame (Mapper.py:498-512), a fresh Adam
        (Mapper.py:570) and clean gradient tables.  row_mask (uint8 [N], 1 on the rows of row_index) lets the backward
        skip the scatter into rows nobody optimises."""
        self.rows = row_index
        self.gs.row_mask = row_mask
        self.adam = optim.Adam(self.eng)
        self.gs.zero_()

    def begin_frame(self):
        """Gradient tables start from zero; rows outside `row_index` may collect (never consumed) scatter
        contributions during the frame, so clear the full tables once per optimize_map call."""
        self.gs.zero_()


class TrackOptimizer:
    """Pose optimisation of one frame: Adam over the 7-vector (quaternion wxyz, translation)."""

    def __init__(self, eng, cfg, dec, knn, pos, geo_feats, col_feats, R, cam_lr, separate_lr=True, w_color=0.5,
                 use_color=True, dynamic_radius=False, dist=None):
        self.eng, self.cfg, self.dec, self.knn = eng, cfg, dec, knn
        self.pos, self.geo, self.col = pos, geo_feats, col_feats
        self.R, self.cam_lr, self.separate_lr = R, cam_lr, separate_lr
        self.w_color, self.use_color = w_color, use_color
        self.batch = RayBatch(eng, R, dynamic_radius)
        self.st = core.RenderState(eng, R, cfg.S, need_act=True)
        self.gs = core.GradState(eng, geo_feats.shape[0], R, dec.n, feats=False, weights=False, rays=True)
        self.g_cam = eng.zeros(7)
        self.eye = None
        self.dist = dist                        # ray-sharded tracking: the 7 pose gradients are summed over ranks

    def track(self, cam7_init, depth_img, color_img, iters, window, intr, rnd_all, r2_map=None, exposure=None):
        """Tracker.run loop body for one frame (Tracker.py:313-401).  cam7_init: [7] device tensor.
        rnd_all int32 [iters, R].  Returns (best cam7, loss log [iters,4]) — one host sync at the end."""
        eng, b, st, gs = self.eng, self.batch, self.st, self.gs
        H, W = depth_img.shape
        cam = cam7_init.clone().contiguous()
        adam = optim.Adam(eng)                  # fresh optimiser per frame (Tracker.py:352)
        # exposure = (mlp_exposure, exposure_feat of this frame): the colour decoder applies sigmoid(rgb @ rot + trans) per
        # sample (decoder.py:534-540); feature and MLP get their own Adam groups at lr 1e-3 (Tracker.py:329-344)
        exp_opt = None
        if exposure is not None:
            exp_opt = torch.optim.Adam([{'params': [exposure[1]], 'lr': 0.001}, {'params': list(exposure[0].parameters()), 'lr': 0.001}])
            if gs.g_affine is None:
                gs.g_affine = eng.zeros(12)
        log = eng.zeros(iters, 4)
        hist = eng.empty(iters, 7)
        if self.eye is None:
            self.eye = torch.eye(4, device=eng.device).reshape(1, 4, 4).contiguous()
        dstack, cstack = depth_img.reshape(1, H, W), color_img.reshape(1, H, W, 3)
        r2s = r2_map.reshape(1, H, W) if r2_map is not None else None
        for it in range(iters):
            hist[it].copy_(cam)
            # pixels, depth, colour (identity pose: only the image gathers are used), then rays of the CURRENT pose
            optim.gather_rays(eng, dstack, cstack, self.eye, None, rnd_all[it], H, W, window, intr, b.as_out(), r2s)
            optim.rays_from_pose(eng, cam, b.pix_i, b.pix_j, intr, b.rays_o, b.rays_d)
            optim.inside_mask(eng, b.gt_depth, None, b.thr, b.scratch_u32, depth_filtered=b.gt_depth)
            aff = None
            if exposure is not None:
                aff_t = exposure[0](exposure[1])                                 # [12], torch autograd through the tiny MLP
                aff = aff_t.detach().float().contiguous()
                gs.g_affine.zero_()
            core.render_forward(eng, self.cfg, st, b.rays_o, b.rays_d, b.gt_depth, self.knn, self.pos, self.geo, self.col,
                                self.dec, 'color', tracker=True, r2_ray=b.r2_ray, save_act=True, affine=aff,
                                extra_flags=_ffi.FLAG_ZERO_ABSENT)
            optim.loss_tracker(eng, st, b.gt_depth, b.gt_color, self.w_color, self.use_color, b.d_depth, b.d_color,
                               log[it], b.loss_scratch)
            core.render_backward(eng, st, gs, b.d_depth, b.d_color)
            optim.pose_bwd(eng, cam, b.pix_i, b.pix_j, intr, gs.g_rays_o, gs.g_rays_d, self.g_cam)
            if exp_opt is not None:
                exp_opt.zero_grad()
                aff_t.backward(gs.g_affine.to(aff_t.dtype))
                exp_opt.step()
            if self.dist is not None:
                self.dist.all_reduce_vec(self.g_cam)
            if self.separate_lr:                # T: lr, quaternion: 0.2*lr (Tracker.py:317-333)
                segs = [('T', cam[4:7], self.g_cam[4:7], self.cam_lr), ('q', cam[0:4], self.g_cam[0:4], 0.2 * self.cam_lr)]
            else:
                segs = [('cam', cam, self.g_cam, self.cam_lr)]
            adam.step(segs)
        best = torch.argmin(log[:, 0])          # Tracker.py:375-377 (first minimum)
        return hist[best].clone(), log
