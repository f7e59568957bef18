"""Thin host layer over the C ABI: owns tensors (PyTorch = device memory + streams), fills
descriptors, launches.  No arithmetic of the hot path happens in Python.

Objects
  Engine          library handle + device + current stream
  KnnIndex        lk_knn_* (replaces NeuralPointCloud's FAISS index, src/neural_point.py:67-72,1659-1708)
  DecoderBlob     packed NICER weights <-> reference state_dict (decoder.py)
  RenderCfg       scalar knobs (configs/point_slam.yaml rendering.*, pointcloud.*, model.*)
  RenderState     caller-owned buffers of one render call (outputs + saved state)
  render_forward  lk_render_fwd  (Renderer.render_batch_ray, src/utils/Renderer.py:71-201)
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _ffi
from ._ffi import RenderDesc, ptr


import os as _os
# LOOPY_CHECK_RANGE=1: every forward tests its fp16-piece operands (LK_FLAG_CHECK_RANGE; a few per cent of the decoder kernels' time)
_CHECK_RANGE = _ffi.FLAG_CHECK_RANGE if _os.environ.get('LOOPY_CHECK_RANGE') == '1' else 0


class Engine:
    """Library + device.  `lib=None` -> the in-tree gfx950 build on cuda:<current> (product path)."""

    def __init__(self, lib=None, device=None):
        self.lib = lib if lib is not None else _ffi.get_lib()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type == 'cuda':
            # the library's side streams take their hardware queues now, before a process group / RCCL creates its own (lk_streams_init)
            with torch.cuda.device(self.device):
                self.lib.check(self.lib.dll.lk_streams_init(), 'lk_streams_init')

    @property
    def stream(self):
        if self.device.type == 'cuda':
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def f32(self, x):
        return torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()

    # ---- operand-range status of the split fp16 products (include/loopy_hip.h: lk_status_peek).  The reference's decoder is plain fp32
    # (src/conv_onet/models/decoder.py:513-546) and has no ceiling; here a weight >= 2^15 or (debug flag) an activation >= 65504 is
    # REPORTED: every later library call fails with LK_ERR_RANGE until clear_status()
    def status(self, sync=False):
        bits = C.c_uint32(0)
        if sync:
            self.lib.check(self.lib.dll.lk_status_sync(self.stream, C.byref(bits)), 'lk_status_sync')
        else:
            self.lib.check(self.lib.dll.lk_status_peek(C.byref(bits)), 'lk_status_peek')
        return int(bits.value)

    def clear_status(self):
        self.lib.check(self.lib.dll.lk_status_clear(), 'lk_status_clear')


@dataclass
class RenderCfg:
    S: int = 5                    # rendering.N_surface
    near_surface: float = 0.98    # rendering.near_end_surface
    far_surface: float = 1.02     # rendering.far_end_surface
    near_end: float = 0.3         # rendering.near_end
    coef: float = 0.1             # rendering.sigmoid_coef_{mapper,tracker}
    min_nn: int = 2               # pointcloud.min_nn_num
    radius_query: float = 0.08    # pointcloud.radius_query
    rel_pos: bool = True          # model.encode_rel_pos_in_col
    exposure: bool = False        # model.encode_exposure
    ray_batch_size: int = 3000    # Renderer(ray_batch_size=...) — far_bb grouping of render_img

    @property
    def r2_static(self):
        return float(np.float32(self.radius_query ** 2))


class KnnIndex:
    """Exact uniform-grid radius-kNN (lk_knn_*)."""

    def __init__(self, eng, capacity, cell_size=0.08, max_cells=1 << 24):
        self.eng = eng
        self.capacity = int(capacity)
        h = C.c_void_p()
        eng.lib.check(eng.lib.dll.lk_knn_create(C.c_float(cell_size), self.capacity, int(max_cells), C.byref(h)),
                      'lk_knn_create')
        self.h = h
        self._pos = None

    def build(self, pos):
        """pos [N,3] fp32 on the engine device; the index keeps a reference."""
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.device == self.eng.device
        self._pos = pos
        self.eng.lib.check(self.eng.lib.dll.lk_knn_build(self.h, ptr(pos), pos.shape[0], self.eng.stream), 'lk_knn_build')

    def append(self, pos_new):
        """lk_knn_append: index M more points behind the existing ones (the handle rebuilds from its own copy)."""
        pos_new = pos_new.contiguous()
        assert pos_new.dtype == torch.float32 and pos_new.device == self.eng.device
        self._keep_new = pos_new
        self.eng.lib.check(self.eng.lib.dll.lk_knn_append(self.h, ptr(pos_new), pos_new.shape[0], self.eng.stream), 'lk_knn_append')

    @property
    def n(self):
        return int(self.eng.lib.dll.lk_knn_size(self.h))

    def query(self, q, r2):
        """q [P,3]; r2 float or [P] tensor -> (d2 [P,8] f32, idx [P,8] i32, count [P] i32)."""
        q = q.contiguous()
        P = q.shape[0]
        d2 = self.eng.empty(P, _ffi.LK_K)
        idx = self.eng.empty(P, _ffi.LK_K, dtype=torch.int32)
        cnt = self.eng.empty(P, dtype=torch.int32)
        per = r2.contiguous() if torch.is_tensor(r2) else None
        self.eng.lib.check(self.eng.lib.dll.lk_knn_query(self.h, ptr(q), P, C.c_float(0.0 if per is not None else r2),
                                                         ptr(per), ptr(d2), ptr(idx), ptr(cnt), self.eng.stream),
                           'lk_knn_query')
        return d2, idx, cnt

    def close(self):
        if self.h:
            self.eng.lib.dll.lk_knn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecoderBlob:
    """The packed fp32 weight blob the kernels read, plus (un)packing to the reference's
    state_dict names.  `blob` is the master copy that Adam updates in place."""

    def __init__(self, eng):
        self.eng = eng
        self.layout = {e['name']: e for e in eng.lib.weight_layout()}
        self.n = eng.lib.blob_floats()
        self.blob = eng.zeros(self.n)
        self.frag = eng.zeros(int(eng.lib.dll.lk_weight_frag_floats()))

    def _view(self, e, flat):
        if e['cols'] == 1 and e['ld'] == 1:
            return flat[e['offset']:e['offset'] + e['rows']]
        return flat[e['offset']:e['offset'] + e['rows'] * e['ld']].view(e['rows'], e['ld'])

    def pack(self, W):
        """W: dict name -> tensor with the reference's shapes (state_dict + 'color_decoder.embedder._B')."""
        host = torch.zeros(self.n, dtype=torch.float32)
        for name, e in self.layout.items():
            if name not in W:
                continue
            t = torch.as_tensor(W[name]).detach().float().cpu()
            v = self._view(e, host)
            if e['cols'] == 1 and e['ld'] == 1:
                v.copy_(t.reshape(-1))
            else:
                t = t.reshape(e['rows'], e['cols'])
                if e['col_split'] > 0:
                    v[:, :e['col_split']] = t[:, :e['col_split']]
                    v[:, e['col_shift']:e['col_shift'] + e['cols'] - e['col_split']] = t[:, e['col_split']:]
                else:
                    v[:, :e['cols']] = t
        self.blob.copy_(host.to(self.eng.device))
        return self.repack(checked=True)

    def repack(self, checked=False):
        """Refresh the fragment copy the kernels read (call after every change of `blob`).
        checked: weights entering from outside (pack = checkpoint load / construction) - waits for the repack and raises LoopyError if a
        matrix entry is non-finite or |w| >= 32768, the ceiling of the forward's split fp16 products; the per-step repacks stay
        asynchronous and report through the sticky status word (Engine.status)."""
        fn = self.eng.lib.dll.lk_weights_repack_checked if checked else self.eng.lib.dll.lk_weights_repack
        self.eng.lib.check(fn(ptr(self.blob), ptr(self.frag), self.eng.stream), 'lk_weights_repack')
        return self

    def unpack(self, flat=None):
        """blob (or a blob-shaped gradient) -> dict of reference-shaped tensors (on CPU)."""
        host = (self.blob if flat is None else flat).detach().cpu()
        out = {}
        for name, e in self.layout.items():
            v = self._view(e, host)
            if e['cols'] == 1 and e['ld'] == 1:
                t = v.clone()
                if name.endswith('output_linear.bias') or name.endswith('.bias'):
                    t = t.reshape(e['rows'])
            elif e['col_split'] > 0:
                t = torch.cat([v[:, :e['col_split']],
                               v[:, e['col_shift']:e['col_shift'] + e['cols'] - e['col_split']]], dim=1)
            else:
                t = v[:, :e['cols']].clone()
            out[name] = t
        return out

    def segment(self, name):
        e = self.layout[name]
        n = e['rows'] * e['ld']
        return e['offset'], n

    def param_ranges(self, names, bridge_padding=False):
        """Merge the blob ranges of `names` into maximal contiguous (offset, length) runs.
        bridge_padding: two ranges separated by alignment padding only (no other tensor of the layout starts in the gap) become one -
        the padding floats of the blob and of its gradient are zero and stay zero under Adam."""
        rng = sorted(self.segment(n) for n in names)
        others = sorted(self.segment(n)[0] for n in self.layout if n not in names)
        out = []
        for off, n in rng:
            end = out[-1][0] + out[-1][1] if out else None
            if out and (end == off or (bridge_padding and end < off and not any(end <= o < off for o in others))):
                out[-1] = (out[-1][0], off + n - out[-1][0])
            else:
                out.append((off, n))
        return out


class RenderState:
    """Buffers of one render call; reused across calls of the same (R, S)."""

    def __init__(self, eng, R, S, need_act=False, flags_for_sizes=0):
        self.eng, self.R, self.S = eng, R, S
        P = R * S
        e = eng
        self.depth, self.var = e.empty(R), e.empty(R)
        self.color = e.empty(R, 3)
        self.valid_ray = e.empty(R, dtype=torch.uint8)
        self.z = e.empty(R, S)
        self.nbr_idx = e.empty(P, 8, dtype=torch.int32)
        self.nbr_w = e.empty(P, 8)
        self.nbr_count = e.empty(P, dtype=torch.int32)
        self.c_geo, self.c_col = e.empty(P, 32), e.empty(P, 32)
        self.raw = e.empty(P, 4)
        self.far_stats = e.empty(max(1, R))
        self.act = None
        if need_act:
            n = int(eng.lib.dll.lk_render_act_floats(R, S, flags_for_sizes))
            self.act = e.empty(max(1, n))
        self.desc = None
        self.keep = None


def fill_desc(eng, cfg, st, rays_o, rays_d, gt_depth, knn, pos, geo_feats, col_feats, dec, stage,
              tracker=False, r2_ray=None, noise_geo=None, noise_col=None, affine=None,
              color_logits=False, save_act=False, stats_chunk=None, extra_flags=0, mapper_loss=None, z_given=None):
    d = RenderDesc()
    R = rays_o.shape[0]
    flags = extra_flags | _CHECK_RANGE
    if stage == 'color':
        flags |= _ffi.FLAG_STAGE_COLOR
    if tracker:
        flags |= _ffi.FLAG_TRACKER
    if cfg.rel_pos:
        flags |= _ffi.FLAG_REL_POS
    if color_logits:
        flags |= _ffi.FLAG_COLOR_LOGITS
    if save_act:
        flags |= _ffi.FLAG_SAVE_ACT
    if geo_feats.dtype == torch.float16:             # opt-in half feature tables: both tables have the same format
        assert col_feats is None or col_feats.dtype == torch.float16
        flags |= _ffi.FLAG_FEATS_F16
    if z_given is not None:          # [R,S] sample depths for the rays without a depth reading (rows of the others are ignored)
        st.z.copy_(z_given.reshape(st.z.shape))
        flags |= _ffi.FLAG_Z_GIVEN
    d.R, d.S, d.flags = R, cfg.S, flags
    d.stats_chunk = int(stats_chunk) if stats_chunk else max(1, R)
    d.rays_o, d.rays_d, d.gt_depth, d.r2_ray = ptr(rays_o), ptr(rays_d), ptr(gt_depth), ptr(r2_ray)
    d.knn = knn.h
    d.pos, d.geo_feats, d.col_feats = ptr(pos), ptr(geo_feats), ptr(col_feats)
    d.weights, d.weights_frag = ptr(dec.blob), ptr(dec.frag)
    d.affine, d.noise_geo, d.noise_col = ptr(affine), ptr(noise_geo), ptr(noise_col)
    d.near_surface, d.far_surface, d.near_end, d.coef = cfg.near_surface, cfg.far_surface, cfg.near_end, cfg.coef
    d.r2_static, d.min_nn = cfg.r2_static, cfg.min_nn
    d.depth, d.var, d.color, d.valid_ray = ptr(st.depth), ptr(st.var), ptr(st.color), ptr(st.valid_ray)
    d.z, d.nbr_idx, d.nbr_w, d.nbr_count = ptr(st.z), ptr(st.nbr_idx), ptr(st.nbr_w), ptr(st.nbr_count)
    d.c_geo, d.c_col, d.raw, d.far_stats, d.act = ptr(st.c_geo), ptr(st.c_col), ptr(st.raw), ptr(st.far_stats), ptr(st.act)
    if mapper_loss is not None:     # (gt_color, w_color, d_depth, d_color, out4): lk_loss_mapper fused into the composite kernel
        gt_color, w_color, d_depth, d_color, out4 = mapper_loss
        d.flags |= _ffi.FLAG_MAPPER_LOSS
        d.loss_gt_color, d.loss_w_color, d.loss_out4 = ptr(gt_color), float(w_color), ptr(out4)
        d.d_depth, d.d_color = ptr(d_depth), ptr(d_color)
        st.keep_loss = mapper_loss
    st.desc = d
    # keep every tensor referenced by raw pointers alive until the next call
    st.keep = (rays_o, rays_d, gt_depth, r2_ray, pos, geo_feats, col_feats, dec, affine, noise_geo, noise_col, knn)      # knn: the backward uses the handle too
    return d


def render_forward(eng, cfg, st, rays_o, rays_d, gt_depth, knn, pos, geo_feats, col_feats, dec, stage, **kw):
    """Fill `st` (depth, var, color, valid_ray + saved state).  All tensors fp32 contiguous on eng.device."""
    d = fill_desc(eng, cfg, st, rays_o, rays_d, gt_depth, knn, pos, geo_feats, col_feats, dec, stage, **kw)
    eng.lib.check(eng.lib.dll.lk_render_fwd(C.byref(d), eng.stream), 'lk_render_fwd')
    return st


class GradState:
    """Gradient buffers of one render call (accumulated into by lk_render_bwd: zero them per step)."""

    def __init__(self, eng, N, R, blob_floats, feats=True, weights=True, rays=False, affine=False):
        self.g_geo = eng.zeros(N, 32) if feats else None
        self.g_col = eng.zeros(N, 32) if feats else None
        self.g_weights = eng.zeros(blob_floats) if weights else None
        self.g_rays_o = eng.zeros(R, 3) if rays else None
        self.g_rays_d = eng.zeros(R, 3) if rays else None
        self.g_affine = eng.zeros(12) if affine else None
        self.row_mask = None            # uint8 [N]: restrict the feature-row gradients to these rows (frustum selection)
        self.geo_decoder = False        # True: g_weights also receives the geometry decoder's matrices / biases (fix_geo_decoder: False)
        self.scratch = None

    def zero_(self):
        for t in (self.g_geo, self.g_col, self.g_weights, self.g_affine):
            if t is not None:
                t.zero_()


def render_backward(eng, st, gs, d_depth, d_color=None, d_var=None):
    """lk_render_bwd on the state of the preceding render_forward(save_act=True).
    Which gradients are produced is decided by the buffers present in `gs`."""
    d = st.desc
    assert d is not None and (d.flags & _ffi.FLAG_SAVE_ACT), 'run render_forward(..., save_act=True) first'
    flags = d.flags & ~(_ffi.FLAG_GRAD_FEATS | _ffi.FLAG_GRAD_WEIGHTS | _ffi.FLAG_GRAD_RAYS | _ffi.FLAG_GRAD_GEO_DECODER)
    if gs.g_geo is not None:
        flags |= _ffi.FLAG_GRAD_FEATS
    if gs.g_weights is not None:
        flags |= _ffi.FLAG_GRAD_WEIGHTS | (_ffi.FLAG_GRAD_GEO_DECODER if gs.geo_decoder else 0)
    if gs.g_rays_o is not None:
        flags |= _ffi.FLAG_GRAD_RAYS
    d.flags = flags
    need = int(eng.lib.dll.lk_render_bwd_scratch_floats(d.R, d.S, flags))
    if gs.scratch is None or gs.scratch.numel() < need:
        gs.scratch = eng.empty(max(1, need))
    d.d_depth, d.d_var, d.d_color = ptr(d_depth), ptr(d_var), ptr(d_color)
    d.g_geo_feats, d.g_col_feats, d.g_weights = ptr(gs.g_geo), ptr(gs.g_col), ptr(gs.g_weights)
    d.g_rays_o, d.g_rays_d, d.g_affine = ptr(gs.g_rays_o), ptr(gs.g_rays_d), ptr(gs.g_affine)
    d.grad_row_mask = ptr(gs.row_mask)
    d.bwd_scratch, d.bwd_scratch_cap = ptr(gs.scratch), gs.scratch.numel()
    st.keep_bwd = (d_depth, d_color, d_var)
    eng.lib.check(eng.lib.dll.lk_render_bwd(C.byref(d), eng.stream), 'lk_render_bwd')
    return gs
