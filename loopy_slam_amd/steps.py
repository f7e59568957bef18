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
import os

import torch

from . import _ffi, core, optim
from ._ffi import ptr

GEO_DECODER_PARAMS = ('geo_decoder.embedder._B',)      # fix_geo_decoder: True (Mapper.py:537-541)
# fix_geo_decoder: False (Mapper.py:524-526): geo_decoder.parameters() - of those the ones the forward uses (the module also owns an
# unused mlp_col_neighbor / embedder_rel_pos, which never receive a gradient: torch's Adam skips them)
GEO_DECODER_ALL_PARAMS = tuple(
    ['geo_decoder.embedder._B'] +
    [f'geo_decoder.pts_linears.{i}.{w}' for i in range(5) for w in ('weight', 'bias')] +
    [f'geo_decoder.fc_c.{i}.{w}' for i in range(5) for w in ('weight', 'bias')] +
    ['geo_decoder.output_linear.weight', 'geo_decoder.output_linear.bias'])
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


class ExposureState:
    """model.encode_exposure (ScanNet): mlp_exposure (Linear 8->128, Softplus(100), Linear 128->12; decoder.py:534-540) and
    the exposure features it is applied to - the tracker's frame (Tracker.py:329-344) or the keyframes of the mapping
    window (Mapper.py:588-607) - evaluated, differentiated and stepped by kernels: lk_exposure_fwd / lk_exposure_bwd + up to five
    lk_adam_step segments (which of them step, and at which rate: adam_segs).  The torch module's parameters are updated IN PLACE (they stay the
    owner for state_dict); several features are stacked into one [F,8] buffer and written back by finish()."""

    LR = 0.001

    def __init__(self, eng, mlp, feats):
        self.eng = eng
        self.W1, self.b1, self.W2, self.b2 = (mlp[0].weight.data, mlp[0].bias.data, mlp[2].weight.data, mlp[2].bias.data)
        for t in (self.W1, self.b1, self.W2, self.b2):
            assert t.is_contiguous() and t.dtype == torch.float32
        assert tuple(self.W1.shape) == (128, 8) and tuple(self.W2.shape) == (12, 128)
        self.single = torch.is_tensor(feats)
        self.orig = [feats] if self.single else list(feats)
        self.F = len(self.orig)
        assert 1 <= self.F <= _ffi.EXPOSURE_MAX_F
        self.feats = feats.data if self.single else torch.stack([f.detach() for f in self.orig]).contiguous()
        self.aff, self.hid = eng.zeros(self.F, 12), eng.zeros(self.F, 128)
        self.g_aff = eng.zeros(self.F, 12)
        self.g = eng.zeros(_ffi.EXPOSURE_GRAD_FLOATS)
        self._desc = None

    def desc(self, lr_mlp, lr_feat=None, only_last_feature=False):
        """lk_exposure_desc for the native loops (lk_track_frame / lk_map_frame): a fresh Adam state in the layout of `g`.
        lr_mlp None: the MLP is frozen."""
        d = _ffi.ExposureDesc()
        d.feats, d.W1, d.b1, d.W2, d.b2, d.F = ptr(self.feats), ptr(self.W1), ptr(self.b1), ptr(self.W2), ptr(self.b2), self.F
        d.aff, d.hid, d.g_aff, d.g = ptr(self.aff), ptr(self.hid), ptr(self.g_aff), ptr(self.g)
        self.adam_mv = self.eng.zeros(2 * _ffi.EXPOSURE_GRAD_FLOATS)
        d.adam = ptr(self.adam_mv)
        d.lr_mlp = -1.0 if lr_mlp is None else float(lr_mlp)
        d.lr_feat = self.LR if lr_feat is None else float(lr_feat)
        d.feat_first, d.feat_count = (self.F - 1, 1) if only_last_feature else (0, self.F)
        self.bwd_scale = self.eng.zeros(1) + 1.0         # kept current by every forward of the MLP inside the loops
        d.bwd_scale = ptr(self.bwd_scale)
        self._desc = d
        return d

    def forward(self):
        e = self.eng
        e.lib.check(e.lib.dll.lk_exposure_fwd(ptr(self.feats), ptr(self.W1), ptr(self.b1), ptr(self.W2), ptr(self.b2), self.F,
                                              ptr(self.aff), ptr(self.hid), e.stream), 'lk_exposure_fwd')
        return self.aff

    def backward(self, g_aff):
        e = self.eng
        e.lib.check(e.lib.dll.lk_exposure_bwd(ptr(self.feats), ptr(self.W1), ptr(self.W2), ptr(self.hid), ptr(g_aff), self.F,
                                              ptr(self.g), e.stream), 'lk_exposure_bwd')

    def adam_segs(self, mlp_lr=None, only_last_feature=False):
        """Tracker (Tracker.py:329-344): the frame's feature and mlp_exposure both at lr 1e-3 (defaults).
        Mapper (Mapper.py:524-570): mlp_exposure is part of color_decoder.parameters() - it steps at the stage's decoders_lr
        (mlp_lr; None = frozen with fix_color_decoder) - and ONLY the current frame's feature (the last one of the window) is an
        Adam parameter (lr 1e-3); the keyframes' features are constants."""
        g, lr = self.g, self.LR
        segs = []
        if mlp_lr is not False:
            ml = lr if mlp_lr is None else mlp_lr
            segs += [('xW1', self.W1.view(-1), g[0:1024], ml), ('xb1', self.b1, g[1024:1152], ml), ('xW2', self.W2.view(-1), g[1152:2688], ml),
                     ('xb2', self.b2, g[2688:2700], ml)]
        if only_last_feature:
            k = self.F - 1
            segs.append(('xf', self.feats.view(-1)[8 * k:8 * k + 8], g[2700 + 8 * k:2700 + 8 * k + 8], lr))
        else:
            segs.append(('xf', self.feats.view(-1), g[2700:2700 + self.F * 8], lr))
        return segs

    def finish(self):
        if not self.single:
            with torch.no_grad():
                for k, t in enumerate(self.orig):
                    t.copy_(self.feats[k])


class MapOptimizer:
    """One optimize_map call: Adam over {decoder params, selected geo rows, selected colour rows}."""

    _tokens = 0                 # process-wide: every constructor and every new_frame takes the next one (caches key on it, never on addresses)

    @classmethod
    def _next_token(cls):
        cls._tokens += 1
        return cls._tokens

    def __init__(self, eng, cfg, dec, knn, pos, geo_feats, col_feats, row_index, R, lrs, w_color=0.1,
                 dynamic_radius=False, fix_color_decoder=False, dist=None, exposure=None, fix_geo_decoder=True):
        """row_index: int32 [n_f] rows being optimised (frustum selection, Mapper.py:498-512) or None = all rows.
        lrs: dict stage -> (decoders_lr, geometry_lr, color_lr)  (configs mapping.stage.*)."""
        self.eng, self.cfg, self.dec, self.knn = eng, cfg, dec, knn
        self.pos, self.geo, self.col = pos, geo_feats, col_feats
        self.rows = row_index
        self.call_token = self._next_token()
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
        self.fix_color_decoder = fix_color_decoder
        # fix_geo_decoder False: the geometry decoder's own matrices join the decoder group (one more backward launch, per-statement path)
        self.fix_geo_decoder = fix_geo_decoder
        self.gs.geo_decoder = not fix_geo_decoder
        self.geo_dec_ranges = dec.param_ranges(list(GEO_DECODER_PARAMS if fix_geo_decoder else GEO_DECODER_ALL_PARAMS),
                                               bridge_padding=not fix_geo_decoder)
        self.col_dec_ranges = dec.param_ranges(cnames)
        self.loss_log = None
        self.dist = dist
        self.it = 0
        self.native_loop = True                 # lk_map_frame; False = one launch sequence per statement (iterate)
        # a call of more iterations than this is issued as consecutive segments (lk_map_desc::it_offset): the work buffer holds one
        # segment's batches and neighbour lists (26 floats per sample and iteration) instead of the whole call's - 3 GB for the 600
        # iterations x 10 000 rays of a ScanNet refinement call.  Single-process loop only (a data-parallel caller walks iteration by iteration).
        # 384: the 300-iteration calls of a mapped frame stay in one piece - every further segment pays one fill of the look-ahead pipeline
        # (measured at 128: 75.7 instead of 74.65 ms per mapped Replica frame, three alternating pairs)
        self.max_call_iters = int(os.environ.get('LOOPY_MAX_CALL_ITERS', '384'))
        self._work = None
        self._nat, self._nat_dirty = None, False    # Adam state of the native loop: [4][n_rows*32] rows, [2][blob] decoders
        self._prepared = None                       # (n_iters, draws, log, work) of a prepare() that run() may build on
        # exposure = (mlp_exposure torch module, [exposure_feat tensor per frame of the window]) for model.encode_exposure
        # (ScanNet): per-keyframe colour affine applied to the RENDERED colour logits (Mapper.py:697-715)
        self.exposure = ExposureState(eng, exposure[0], exposure[1]) if exposure is not None else None
        self.ba = None

    def enable_ba(self, cams7, trainable, lr_of_iteration):
        """mapping.BA (Mapper.py:541-566): the poses of the window's frames join the optimiser as unnormalised-quaternion 7-vectors
        (one Adam group, lr BA_cam_lr inside the schedule's window and 0 outside, Mapper.py:602-607), the batch is rendered with
        is_tracker=True (Mapper.py:685: the interpolation weights become functions of the sample positions) and the rays of frame f
        are rebuilt from cams7[f] in every iteration (Mapper.py:629-643).  The rays of a batch are grouped by frame (frame_id
        ascending, equal counts).
        cams7: [F,7] device tensor (stepped in place); trainable: F bools - the oldest keyframe stays fixed (Mapper.py:547-548) and
        keeps the rays of its pose MATRIX; lr_of_iteration: it -> lr of the camera group.  Per-statement path only."""
        eng = self.eng
        F = cams7.shape[0]
        assert cams7.is_contiguous() and cams7.shape == (F, 7) and len(trainable) == F and self.R % F == 0
        self.ba = dict(cams=cams7, g=eng.zeros(F, 7), trainable=[bool(t) for t in trainable], pix=self.R // F, lr=lr_of_iteration)
        if self.gs.g_rays_o is None:
            self.gs.g_rays_o, self.gs.g_rays_d = eng.zeros(self.R, 3), eng.zeros(self.R, 3)

    def iterate(self, stage, frames, rnd, frame_id, window, intr, H, W, log_row=None):
        """One joint iteration (Mapper.py:576-735).
        frames: (depth_stack [F,H,W], color_stack [F,H,W,3], c2w_stack [F,4,4], r2_map_stack or None);
        rnd int32 [R] window-pixel draws, frame_id int32 [R]."""
        eng, b, st, gs = self.eng, self.batch, self.st, self.gs
        depth_stack, color_stack, c2w_stack, r2_stack = frames
        optim.gather_rays(eng, depth_stack, color_stack, c2w_stack, frame_id, rnd, H, W, window, intr, b.as_out(), r2_stack)
        optim.inside_mask(eng, b.gt_depth, None, b.thr, b.scratch_u32, depth_filtered=b.gt_depth)
        ba = self.ba
        if ba is not None:              # rays of the trainable frames from their 7-vectors (get_camera_from_tensor, Mapper.py:629-643)
            n = ba['pix']
            for f, tr in enumerate(ba['trainable']):
                if tr:
                    sl = slice(f * n, (f + 1) * n)
                    optim.rays_from_pose(eng, ba['cams'][f], b.pix_i[sl], b.pix_j[sl], intr, b.rays_o[sl], b.rays_d[sl])
        out4 = log_row if log_row is not None else self._out4()
        xs = self.exposure if stage == 'color' else None
        # without exposure encoding the mapper loss (Mapper.py:691-720) is evaluated inside the composite kernel
        fused = None if xs is not None else (b.gt_color, self.w_color, b.d_depth, b.d_color, out4)
        core.render_forward(eng, self.cfg, st, b.rays_o, b.rays_d, b.gt_depth, self.knn, self.pos, self.geo, self.col,
                            self.dec, stage, r2_ray=b.r2_ray, save_act=True, tracker=ba is not None,
                            # L1 sums: |d depth| = 1, |d colour| = w - unit scale, the backward may use fp16 pieces.  NOT with exposure
                            # encoding: there d logits = w sigma' A with a LEARNED 3x3 A per keyframe, nothing bounds it
                            extra_flags=_ffi.FLAG_ZERO_ABSENT | (_ffi.FLAG_UNIT_LOSS_GRADS if self.exposure is None else 0),
                            color_logits=self.exposure is not None, mapper_loss=fused)
        if xs is not None:
            # the renderer returned colour LOGITS; the rays of keyframe f get sigmoid(logits @ rot_f + trans_f)
            # (Mapper.py:697-715): d depth / d logits for the backward, d loss / d affine for the exposure MLP
            xs.forward()
            eng.lib.check(eng.lib.dll.lk_loss_mapper_exposure(self.R, ptr(st.depth), ptr(st.color), ptr(st.valid_ray), ptr(b.gt_depth),
                                                              ptr(b.gt_color), ptr(frame_id), ptr(xs.aff), xs.F,
                                                              _ffi.C.c_float(self.w_color), ptr(b.d_depth), ptr(b.d_color), ptr(out4),
                                                              ptr(xs.g_aff), eng.stream), 'lk_loss_mapper_exposure')
        core.render_backward(eng, st, gs, b.d_depth, b.d_color)
        if ba is not None:              # d rays -> d pose of every trainable frame (rows of the fixed frame stay zero: Adam leaves it)
            n = ba['pix']
            for f, tr in enumerate(ba['trainable']):
                if tr:
                    sl = slice(f * n, (f + 1) * n)
                    optim.pose_bwd(eng, ba['cams'][f], b.pix_i[sl], b.pix_j[sl], intr, gs.g_rays_o[sl], gs.g_rays_d[sl], ba['g'][f])
        if self.dist is not None:
            self.dist.all_reduce_grads(self, stage, it=None)
            if ba is not None:
                self.dist.all_reduce_vec(ba['g'])
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
        if xs is not None:
            xs.backward(xs.g_aff)
            segs += xs.adam_segs(mlp_lr=False if self.fix_color_decoder else dlr, only_last_feature=True)
        if ba is not None:              # one group for all camera tensors (Mapper.py:565-566); elementwise, so one segment
            segs.append(('cams', ba['cams'].view(-1), ba['g'].view(-1), float(ba['lr'](self.it))))
        self.adam.step(segs, zero_grad=True)
        if stage == 'color' or not self.fix_geo_decoder:
            # the geometry stage only moves the embedding matrices (read from the plain blob); the MFMA fragments are
            # copies of the colour-decoder matrices, which only change in the colour stage (and of the geometry decoder's, which
            # only change without fix_geo_decoder)
            self.dec.repack()
        self.it += 1
        return out4

    def run(self, n_iters, n_geo_iters, frames, rnd_all, frame_id, window, intr, H, W, log):
        """All joint iterations of one optimize_map call (Mapper.py:576-735): iteration it runs stage 'geometry' iff
        it < n_geo_iters.  rnd_all int32 [n_iters, R]; log [n_iters, 4].  Without exposure encoding this is lk_map_frame - one
        C-ABI call for the whole loop single-GPU, two calls per iteration around the gradient all-reduce multi-GPU; with
        exposure encoding the per-statement path (iterate)."""
        if not self._takes_native_loop():
            for it in range(n_iters):
                self.iterate('geometry' if it < n_geo_iters else 'color', frames, rnd_all[it], frame_id, window, intr, H, W, log_row=log[it])
            return log
        eng = self.eng
        d, seg_iters = self._native_desc(n_iters, n_geo_iters, frames, rnd_all, frame_id, window, intr, H, W, log)
        C = _ffi.C
        dll = eng.lib.dll
        prepared, self._prepared = self._prepared, None
        if self.dist is None:
            for s0 in range(0, n_iters, seg_iters):         # (one pass unless the call is longer than max_call_iters)
                n = min(seg_iters, n_iters - s0)
                d.it_offset, d.iters = s0, n
                d.rnd, d.log = ptr(rnd_all[s0:]), ptr(log[s0:])
                # the first segment's batches may have been assembled ahead of the row selection (prepare)
                d.batches_ready = 1 if (s0 == 0 and prepared == self._prepare_key(n_iters, frames, rnd_all, frame_id, window, intr, H, W, log)) else 0
                eng.lib.check(dll.lk_map_frame(C.byref(d), 0, n, 3, eng.stream), 'lk_map_frame')
        else:
            self._nat_desc = d
            d.signal_rows = 1 if self.dist.overlaps_rows(self) else 0
            for it in range(n_iters):
                eng.lib.check(dll.lk_map_frame(C.byref(d), it, it + 1, 1, eng.stream), 'lk_map_frame')
                if self.rows is None and it + 1 < n_iters and self._nat_lists is not None:
                    self.dist.prefetch_touched(self, it + 1)        # next iteration's row list, agreed beside this iteration's render
                self.dist.all_reduce_grads(self, 'geometry' if it < n_geo_iters else 'color', it=it, desc=d)
                if self.rows is None:
                    self.dist.flag_union(self)      # rows touched by any rank: what the step of a whole-map iteration visits
                eng.lib.check(dll.lk_map_frame(C.byref(d), it, it + 1, 2, eng.stream), 'lk_map_frame')
        self.it += n_iters
        return log

    def _takes_native_loop(self):
        return self.native_loop and self.ba is None and not (self.exposure is not None and self.R > 16384)

    def prepare(self, n_iters, n_geo_iters, frames, rnd_all, frame_id, window, intr, H, W, log):
        """The row-independent head of run(): clears the gradient tables and assembles the batches of the call (lk_map_prepare) - for a
        caller whose row list comes out of a device-side selection with a count read-back (Mapper.get_mask_from_c2w): issued BEFORE the
        read-back, the device works through the fills and the assembly while the host waits and builds the descriptor.  run() with the
        same arguments follows (new_frame(..., zero=False) in between).  No-op on the paths that have no batch assembly."""
        self._prepared = None
        if not self._takes_native_loop() or self.dist is not None or self.R > 16384:
            return False
        self.gs.zero_()
        d, seg_iters = self._native_desc(n_iters, n_geo_iters, frames, rnd_all, frame_id, window, intr, H, W, log, rows_known=False)
        d.iters, d.it_offset = min(seg_iters, n_iters), 0
        self.eng.lib.check(self.eng.lib.dll.lk_map_prepare(_ffi.C.byref(d), self.eng.stream), 'lk_map_prepare')
        self._prepared = self._prepare_key(n_iters, frames, rnd_all, frame_id, window, intr, H, W, log)
        return True

    def _prepare_key(self, n_iters, frames, rnd_all, frame_id, window, intr, H, W, log):
        """What a prepare() assembled its batches FROM: a run() may build on them only if every input is the same buffer with the same
        geometry (the draws, the frame stacks, the frame ids, the window, the intrinsics); prepare() keeps the tensors alive
        (_native_desc: _keep_native), so an address cannot have been freed and handed out again in between."""
        ptrs = tuple(t.data_ptr() if t is not None else 0 for t in (rnd_all, log, frame_id, self._work) + tuple(frames))
        return (n_iters, ptrs, tuple(int(x) for x in window), tuple(float(x) for x in intr), int(H), int(W))

    def _native_desc(self, n_iters, n_geo_iters, frames, rnd_all, frame_id, window, intr, H, W, log, rows_known=True):
        """lk_map_desc of an optimize_map call + the segment length.  rows_known False (prepare): the optimiser-state part is left out."""
        eng, b, st, gs = self.eng, self.batch, self.st, self.gs
        depth_stack, color_stack, c2w_stack, r2_stack = frames
        core.fill_desc(eng, self.cfg, st, b.rays_o, b.rays_d, b.gt_depth, self.knn, self.pos, self.geo, self.col, self.dec, 'color',
                       r2_ray=b.r2_ray, save_act=True)
        # the scratch layout depends on the flags: size it with exactly the words lk_map_frame renders with, in both stages
        unit = _ffi.FLAG_UNIT_LOSS_GRADS if self.exposure is None else 0       # with exposure encoding the loss gradients are not unit scale
        base = (_ffi.FLAG_REL_POS if self.cfg.rel_pos else 0) | unit | _ffi.FLAG_SAVE_ACT | _ffi.FLAG_GRAD_FEATS | \
            _ffi.FLAG_GRAD_WEIGHTS | _ffi.FLAG_ZERO_ABSENT | _ffi.FLAG_MAPPER_LOSS | (0 if self.fix_geo_decoder else _ffi.FLAG_GRAD_GEO_DECODER)
        need = max(int(eng.lib.dll.lk_render_bwd_scratch_floats(self.R, self.cfg.S, base | extra)) for extra in (0, _ffi.FLAG_STAGE_COLOR))
        if gs.scratch is None or gs.scratch.numel() < need:
            gs.scratch = eng.empty(max(1, need))
        d = _ffi.MapDesc()
        C = _ffi.C
        C.memmove(C.byref(d.render), C.byref(st.desc), C.sizeof(_ffi.RenderDesc))
        r = d.render
        r.flags = (_ffi.FLAG_REL_POS if self.cfg.rel_pos else 0) | unit      # L1 sums: |d depth|, |d colour| <= 1
        if self.geo.dtype == torch.float16:
            r.flags |= _ffi.FLAG_FEATS_F16
        r.d_depth, r.d_color = ptr(b.d_depth), ptr(b.d_color)
        r.g_geo_feats, r.g_col_feats, r.g_weights = ptr(gs.g_geo), ptr(gs.g_col), ptr(gs.g_weights)
        r.grad_row_mask, r.bwd_scratch, r.bwd_scratch_cap = ptr(gs.row_mask), ptr(gs.scratch), gs.scratch.numel()
        H0, H1, W0, W1 = window
        d.depth_stack, d.color_stack, d.c2w_stack, d.c2w_stride = ptr(depth_stack), ptr(color_stack), ptr(c2w_stack), c2w_stack.shape[-2] * 4
        d.r2_map_stack, d.frame_id = ptr(r2_stack), ptr(frame_id)
        rnd_all = rnd_all.contiguous()
        assert rnd_all.dtype == torch.int32 and rnd_all.shape[0] >= n_iters and rnd_all.shape[1] == self.R and log.is_contiguous()
        d.rnd = ptr(rnd_all)
        d.H, d.W, d.H0, d.W0, d.w = H, W, H0, W0, W1 - W0
        d.fx, d.fy, d.cx, d.cy = intr
        d.gt_color, d.thr, d.scratch_u32 = ptr(b.gt_color), ptr(b.thr), ptr(b.scratch_u32)
        d.w_color, d.log = self.w_color, ptr(log)
        d.weights_rw, d.weights_frag_rw = ptr(self.dec.blob), ptr(self.dec.frag)
        d.geo_feats_rw, d.col_feats_rw = ptr(self.geo), ptr(self.col)
        n_rows = self.rows.numel() if self.rows is not None else self.geo.shape[0]
        d.rows, d.n_rows = ptr(self.rows), n_rows
        if rows_known:
            if self._nat is None or self._nat[0].numel() != 4 * n_rows * 32:
                self._nat = (eng.zeros(4 * n_rows * 32), eng.zeros(2 * self.dec.n))
            elif self._nat_dirty:
                self._nat[0].zero_(); self._nat[1].zero_()
            self._nat_dirty = True
            d.adam_rows, d.adam_dec = ptr(self._nat[0]), ptr(self._nat[1])
        assert len(self.geo_dec_ranges) <= _ffi.MAX_SPANS and len(self.col_dec_ranges) <= _ffi.MAX_SPANS
        for k, (o, n) in enumerate(self.geo_dec_ranges):
            d.geo_dec[k].offset, d.geo_dec[k].n = o, n
        for k, (o, n) in enumerate(self.col_dec_ranges):
            d.col_dec[k].offset, d.col_dec[k].n = o, n
        d.n_geo_dec, d.n_col_dec = len(self.geo_dec_ranges), len(self.col_dec_ranges)
        for si, stage in enumerate(('geometry', 'color')):
            for k in range(3):
                d.lr[si][k] = self.lrs[stage][k]
        d.iters, d.n_geo_iters = n_iters, n_geo_iters
        d.train_geo_decoder = 0 if self.fix_geo_decoder else 1
        if self.exposure is not None:
            # Mapper.py:524-570: mlp_exposure steps with the colour decoder (decoders_lr of stage 'color', frozen with it), of the window's
            # exposure features only the current frame's (the last) is an Adam parameter
            xd = self.exposure.desc(None if self.fix_color_decoder else self.lrs['color'][0], only_last_feature=True)
            d.exposure = C.pointer(xd)
        seg_iters = n_iters if (self.dist is not None or n_iters <= self.max_call_iters) else self.max_call_iters
        d.iters = seg_iters
        need = int(eng.lib.dll.lk_map_work_floats(self.R, self.cfg.S, seg_iters)) if self.R <= 16384 else 0
        if need and (self._work is None or self._work.numel() < need):
            self._work = eng.empty(need)
        d.work = ptr(self._work) if need else 0
        d.union_rows_flagged = 1 if (self.dist is not None and self.rows is None and self.geo.shape[0] <= self.knn.capacity) else 0
        self._nat_lists = (int(eng.lib.dll.lk_map_work_nbr_idx(self.R, self.cfg.S, n_iters)), n_iters) if need else None
        self._keep_native = (depth_stack, color_stack, c2w_stack, r2_stack, frame_id, rnd_all, log)
        return d, seg_iters

    def nbr_idx_of(self, it=None):
        """Neighbour lists [R*S, 8] (int32) of iteration `it` of the running lk_map_frame call (it keeps the lists of every iteration
        in its work buffer: the search runs ahead of the loop); it = None, or the per-statement path: the lists in the state."""
        if it is not None and getattr(self, '_nat_lists', None) is not None and self.native_loop:
            off, _ = self._nat_lists
            P8 = self.R * self.cfg.S * 8
            return self._work[off + it * P8: off + (it + 1) * P8].view(torch.int32)
        return self.st.nbr_idx

    def wait_lists(self, it):
        """The current stream waits until lk_map_frame's look-ahead search has written iteration `it`'s lists."""
        if getattr(self, '_nat_desc', None) is not None and getattr(self, '_nat_lists', None) is not None:
            self.eng.lib.check(self.eng.lib.dll.lk_map_wait_lists(_ffi.C.byref(self._nat_desc), it, self.eng.stream), 'lk_map_wait_lists')

    def finish(self):
        """End of the optimize_map call: stacked exposure features go back to the keyframes' tensors."""
        if self.exposure is not None:
            self.exposure.finish()

    def _out4(self):
        if self.loss_log is None:
            self.loss_log = self.eng.zeros(4)
        return self.loss_log

    def new_frame(self, row_index, row_mask=None, zero=True):
        """Start of an optimize_map call: the frustum rows of the new frame (Mapper.py:498-512), a fresh Adam
        (Mapper.py:570) and clean gradient tables (zero=False: prepare() has cleared them already).  row_mask (uint8 [N], 1 on the
        rows of row_index) lets the backward skip the scatter into rows nobody optimises."""
        self.rows = row_index
        self.call_token = self._next_token()
        self.gs.row_mask = row_mask
        self.adam = optim.Adam(self.eng)
        if zero:
            self.gs.zero_()
        self._nat_dirty = True                  # fresh optimiser: the native loop's state is re-zeroed (or re-sized) on its next run

    def begin_frame(self):
        """Gradient tables start from zero; rows outside `row_index` may collect (never consumed) scatter
        contributions during the frame, so clear the full tables once per optimize_map call."""
        self.gs.zero_()


class TrackOptimizer:
    """Pose optimisation of one frame: Adam over the 7-vector (quaternion wxyz, translation)."""

    def __init__(self, eng, cfg, dec, knn, pos, geo_feats, col_feats, R, cam_lr, separate_lr=True, w_color=0.5,
                 use_color=True, dynamic_radius=False, dist=None, handle_dynamic=True, shard_rays=False):
        self.eng, self.cfg, self.dec, self.knn = eng, cfg, dec, knn
        self.pos, self.geo, self.col = pos, geo_feats, col_feats
        self.R, self.cam_lr, self.separate_lr = R, cam_lr, separate_lr
        self.w_color, self.use_color = w_color, use_color
        self.handle_dynamic = handle_dynamic    # False: median-of-residual outlier mask (Tracker.py:177-179)
        # shard_rays (opt-in, `tracking.shard_rays`; SURVEY 8(e) "contiguous ray ranges (tracking, ...)"): rank r renders and back-propagates the
        # contiguous range parallel.ray_range(R, r, world) of every iteration's draws, ONE all-reduce per iteration sums {pose gradient (7), loss
        # row (4), exposure-affine gradient (12)} and the identical Adam step runs on every rank.  Per-statement path (the exchange sits between
        # the backward and the step).  The inside mask and the residual median are per shard, as the mapper's (DESIGN section 4 "deviations").
        self.shard_rays = bool(shard_rays and dist is not None and dist.world > 1)
        self.R_own = R
        if self.shard_rays:
            from . import parallel
            self.own = parallel.ray_range(R, dist.rank, dist.world, 1)
            self.R_own = max(1, max(parallel.ray_range(R, r, dist.world, 1)[1] - parallel.ray_range(R, r, dist.world, 1)[0] for r in range(dist.world)))
        self.batch = RayBatch(eng, self.R_own, dynamic_radius)
        self.st = core.RenderState(eng, self.R_own, cfg.S, need_act=True)
        self.gs = core.GradState(eng, geo_feats.shape[0], self.R_own, dec.n, feats=False, weights=False, rays=True)
        self.g_cam = eng.zeros(7)
        self.eye = None
        # Multi-GPU: tracking is REPLICATED, not sharded.  An iteration at the reference's batch (1 500-5 000 rays) sits at the
        # dependent-launch latency floor (~130 us for ~6 us of matrix work): a shard of it is no faster, and a
        # per-iteration all-reduce of the 7 pose gradients (>= 30 us over RCCL) would only add to the chain.  Every rank runs the same
        # native loop on the same draws against its replica of the map; rank 0's result is broadcast once per frame so that the
        # ranks cannot drift apart through the loss log's float atomics (the candidate choice, Tracker.py:375-377).
        self.dist = dist
        self.native_loop = not self.shard_rays  # lk_track_frame (one call per frame); False = one launch sequence per statement

    def track(self, cam7_init, depth_img, color_img, iters, window, intr, rnd_all, r2_map=None, exposure=None):
        """Tracker.run loop body for one frame (Tracker.py:313-401).  cam7_init: [7] device tensor.
        rnd_all int32 [iters, R].  Returns (best cam7, loss log [iters,4]) — one host sync at the end."""
        eng, b, st, gs = self.eng, self.batch, self.st, self.gs
        H, W = depth_img.shape
        cam = cam7_init.clone().contiguous()
        adam = optim.Adam(eng)                  # fresh optimiser per frame (Tracker.py:352)
        # exposure = (mlp_exposure, exposure_feat of this frame): the colour decoder applies sigmoid(rgb @ rot + trans) per
        # sample (decoder.py:534-540); feature and MLP get their own Adam groups at lr 1e-3 (Tracker.py:329-344)
        xs = None
        if exposure is not None:
            xs = ExposureState(eng, exposure[0], exposure[1])
            if gs.g_affine is None:
                gs.g_affine = eng.zeros(12)
        log = eng.zeros(iters, 4)
        hist = eng.empty(iters, 7)
        if self.native_loop:
            # the whole loop as ONE C-ABI call (lk_track_frame): no interpreter between the launches
            self._track_native(cam, depth_img, color_img, iters, window, intr, rnd_all, r2_map, hist, log, xs)
            best = torch.argmin(log[:, 0])      # Tracker.py:375-377 (first minimum)
            return self._agree(hist.index_select(0, best.reshape(1))[0], xs), log   # (hist[best] would read `best` back: a host sync)
        if self.eye is None:
            self.eye = torch.eye(4, device=eng.device).reshape(1, 4, 4).contiguous()
        dstack, cstack = depth_img.reshape(1, H, W), color_img.reshape(1, H, W, 3)
        r2s = r2_map.reshape(1, H, W) if r2_map is not None else None
        xch = eng.zeros(7 + 4 + 12) if self.shard_rays else None
        for it in range(iters):
            if self.separate_lr:
                hist[it].copy_(cam)             # candidate = detached copy of the pose BEFORE the step (Tracker.py:363-377)
            # pixels, depth, colour (identity pose: only the image gathers are used), then rays of the CURRENT pose
            draws = rnd_all[it]
            if self.shard_rays:                 # this rank's contiguous range of the batch, padded to the largest range with its last draw
                lo, hi = self.own
                draws = draws[lo:hi] if hi > lo else draws[:1]
                if draws.shape[0] < self.R_own:
                    draws = torch.cat([draws, draws[-1:].expand(self.R_own - draws.shape[0])])
                draws = draws.contiguous()
            optim.gather_rays(eng, dstack, cstack, self.eye, None, draws, H, W, window, intr, b.as_out(), r2s)
            if self.shard_rays and hi - lo < self.R_own:
                b.gt_depth[max(hi - lo, 0):].zero_()             # the padding: rays without a reading are absent from the losses and carry no gradient
            optim.rays_from_pose(eng, cam, b.pix_i, b.pix_j, intr, b.rays_o, b.rays_d)
            optim.inside_mask(eng, b.gt_depth, None, b.thr, b.scratch_u32, depth_filtered=b.gt_depth)
            aff = None
            if xs is not None:
                aff = xs.forward()[0]                                            # [12] = (rot 3x3 | trans 3) of this frame
                gs.g_affine.zero_()
            core.render_forward(eng, self.cfg, st, b.rays_o, b.rays_d, b.gt_depth, self.knn, self.pos, self.geo, self.col,
                                self.dec, 'color', tracker=True, r2_ray=b.r2_ray, save_act=True, affine=aff,
                                extra_flags=_ffi.FLAG_ZERO_ABSENT)
            optim.loss_tracker(eng, st, b.gt_depth, b.gt_color, self.w_color, self.use_color, b.d_depth, b.d_color,
                               log[it], b.loss_scratch, handle_dynamic=self.handle_dynamic)
            core.render_backward(eng, st, gs, b.d_depth, b.d_color)
            optim.pose_bwd(eng, cam, b.pix_i, b.pix_j, intr, gs.g_rays_o, gs.g_rays_d, self.g_cam)
            if self.shard_rays:                 # the iteration's one exchange: sums over the ranks' ranges
                xch[:7].copy_(self.g_cam); xch[7:11].copy_(log[it])
                if xs is not None:
                    xch[11:].copy_(gs.g_affine)
                self.dist.all_reduce_vec(xch)
                self.g_cam.copy_(xch[:7]); log[it].copy_(xch[7:11])
                if xs is not None:
                    gs.g_affine.copy_(xch[11:])
            if self.separate_lr:                # T: lr, quaternion: 0.2*lr (Tracker.py:317-333)
                segs = [('T', cam[4:7], self.g_cam[4:7], self.cam_lr), ('q', cam[0:4], self.g_cam[0:4], 0.2 * self.cam_lr)]
            else:
                segs = [('cam', cam, self.g_cam, self.cam_lr)]
            if xs is not None:
                xs.backward(gs.g_affine)
                segs = segs + xs.adam_segs()
            adam.step(segs)
            if not self.separate_lr:
                hist[it].copy_(cam)             # one leaf tensor stepped in place: the candidate is the pose AFTER the update
        best = torch.argmin(log[:, 0])          # Tracker.py:375-377 (first minimum)
        return self._agree(hist.index_select(0, best.reshape(1))[0], xs), log

    def _agree(self, cam7, xs=None):
        """Replicated tracking: every rank continues from rank 0's pose - and, with exposure encoding, from rank 0's exposure feature and
        mlp_exposure tensors: the tracking loop steps them on every rank (Tracker.py:329-344) from gradients that were summed with float
        atomics, and the mapper's no-parameter-broadcast scheme needs bit-identical replicas (2 708 floats once per frame)."""
        if self.dist is not None:
            self.dist.broadcast(cam7, src=0)
            if xs is not None:
                for t in (xs.feats, xs.W1, xs.b1, xs.W2, xs.b2):
                    self.dist.broadcast(t, src=0)
        return cam7

    def _track_native(self, cam, depth_img, color_img, iters, window, intr, rnd_all, r2_map, hist, log, xs=None):
        """lk_track_frame: descriptor of the render buffers + the loop's own buffers (all owned here)."""
        eng, b, st, gs = self.eng, self.batch, self.st, self.gs
        H, W = depth_img.shape
        H0, H1, W0, W1 = window
        core.fill_desc(eng, self.cfg, st, b.rays_o, b.rays_d, b.gt_depth, self.knn, self.pos, self.geo, self.col, self.dec, 'color',
                       tracker=True, r2_ray=b.r2_ray, save_act=True, extra_flags=_ffi.FLAG_ZERO_ABSENT)
        flags = st.desc.flags | _ffi.FLAG_GRAD_RAYS
        need = int(eng.lib.dll.lk_render_bwd_scratch_floats(self.R, self.cfg.S, flags))
        if gs.scratch is None or gs.scratch.numel() < need:
            gs.scratch = eng.empty(max(1, need))
        d = _ffi.TrackDesc()
        C = _ffi.C
        C.memmove(C.byref(d.render), C.byref(st.desc), C.sizeof(_ffi.RenderDesc))
        r = d.render
        r.flags = (_ffi.FLAG_REL_POS if self.cfg.rel_pos else 0) | (_ffi.FLAG_FEATS_F16 if self.geo.dtype == torch.float16 else 0)
        r.d_depth, r.d_color = ptr(b.d_depth), ptr(b.d_color)
        r.g_rays_o, r.g_rays_d, r.bwd_scratch, r.bwd_scratch_cap = ptr(gs.g_rays_o), ptr(gs.g_rays_d), ptr(gs.scratch), gs.scratch.numel()
        d.depth_img, d.color_img, d.r2_map = ptr(depth_img), ptr(color_img), ptr(r2_map)
        d.H, d.W, d.H0, d.W0, d.w = H, W, H0, W0, W1 - W0
        d.fx, d.fy, d.cx, d.cy = intr
        rnd_all = rnd_all.contiguous()
        assert rnd_all.dtype == torch.int32 and rnd_all.shape[0] >= iters and rnd_all.shape[1] == self.R
        d.rnd, d.gt_color, d.pix_i, d.pix_j = ptr(rnd_all), ptr(b.gt_color), ptr(b.pix_i), ptr(b.pix_j)
        d.thr, d.scratch_u32, d.loss_scratch = ptr(b.thr), ptr(b.scratch_u32), ptr(b.loss_scratch)
        if getattr(self, 'adam_mv', None) is None:
            self.adam_mv = eng.zeros(14)
        d.cam7, d.g_cam7, d.adam_mv = ptr(cam), ptr(self.g_cam), ptr(self.adam_mv)
        d.lr_T, d.lr_q = self.cam_lr, (0.2 * self.cam_lr if self.separate_lr else self.cam_lr)
        d.w_color, d.use_color, d.hist_post = self.w_color, optim.track_loss_flags(self.use_color, self.handle_dynamic), int(not self.separate_lr)
        d.hist, d.log, d.iters = ptr(hist), ptr(log), iters
        need = int(eng.lib.dll.lk_track_work_floats(self.R, self.cfg.S, iters)) if self.R <= 8192 else 0
        if need and (getattr(self, '_work', None) is None or self._work.numel() < need):
            self._work = eng.empty(need)
        d.work = ptr(self._work) if need else 0
        if xs is not None:              # Tracker.py:329-344: the frame's feature and mlp_exposure, both at lr 1e-3
            d.exposure = C.pointer(xs.desc(ExposureState.LR))
        self._keep_native = (depth_img, color_img, r2_map, rnd_all, cam, hist, log, xs)
        eng.lib.check(eng.lib.dll.lk_track_frame(C.byref(d), eng.stream), 'lk_track_frame')
