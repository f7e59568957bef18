"""Host wrappers of the per-iteration kernels around the render call: fused losses, multi-tensor
Adam, pose<->rays, inside-mask and compaction (lk_loss_*, lk_adam_step, lk_rays_from_pose,
lk_pose_bwd, lk_inside_mask, lk_compact)."""
import ctypes as C

import torch

from . import _ffi
from ._ffi import ptr, AdamSeg


def loss_mapper(eng, st, gt_depth, gt_color, w_color, use_color, d_depth, d_color, out4):
    """Mapper.py:691-720 (non-exposure).  out4 = [loss, geo, colour, #masked] (device)."""
    R = gt_depth.shape[0]
    eng.lib.check(eng.lib.dll.lk_loss_mapper(R, ptr(st.depth), ptr(st.color), ptr(st.valid_ray), ptr(gt_depth),
                                             ptr(gt_color), C.c_float(w_color), int(bool(use_color)),
                                             ptr(d_depth), ptr(d_color), ptr(out4), eng.stream), 'lk_loss_mapper')


TRACK_USE_COLOR, TRACK_MEDIAN_MASK = 1, 2          # include/loopy_hip.h: flag word of lk_loss_tracker / lk_track_desc::use_color


def track_loss_flags(use_color, handle_dynamic=True):
    return (TRACK_USE_COLOR if use_color else 0) | (0 if handle_dynamic else TRACK_MEDIAN_MASK)


def loss_tracker(eng, st, gt_depth, gt_color, w_color, use_color, d_depth, d_color, out4, scratch, handle_dynamic=True):
    """Tracker.py:169-191.  handle_dynamic False: the mask compares |gt - depth| with 10 x its median (Tracker.py:177-179).
    scratch: R+8 floats."""
    R = gt_depth.shape[0]
    eng.lib.check(eng.lib.dll.lk_loss_tracker(R, ptr(st.depth), ptr(st.var), ptr(st.color), ptr(gt_depth), ptr(gt_color),
                                              C.c_float(w_color), track_loss_flags(use_color, handle_dynamic), ptr(d_depth), ptr(d_color),
                                              ptr(out4), ptr(scratch), eng.stream), 'lk_loss_tracker')


class Adam:
    """torch.optim.Adam (amsgrad=False, weight_decay=0) over flat fp32 tensors, one launch per step.
    Each segment is (param, grad) with its own lr and step count; a segment whose lr group has no
    gradient this step is simply not passed (torch skips parameters whose .grad is None)."""

    def __init__(self, eng, beta1=0.9, beta2=0.999, eps=1e-8):
        self.eng, self.beta1, self.beta2, self.eps = eng, beta1, beta2, eps
        self.state = {}

    def step(self, segs, zero_grad=False):
        """segs: list of (key, param, grad, lr) or (key, table, grad_table, lr, row_index).
        Without row_index param/grad are contiguous fp32 tensors of equal numel; with row_index
        (int32 [n_rows]) the rows table[row_index] of a [N, row_len] table are updated in place."""
        assert len(segs) <= _ffi.ADAM_MAX_SEG
        arr = (AdamSeg * max(1, len(segs)))()
        keep = []
        for k, seg in enumerate(segs):
            key, p, g, lr = seg[:4]
            rows = seg[4] if len(seg) > 4 else None
            n = p.numel() if rows is None else rows.numel() * p.shape[1]
            st = self.state.get(key)
            if st is None:
                st = self.state[key] = dict(m=torch.zeros(n, dtype=torch.float32, device=p.device),
                                            v=torch.zeros(n, dtype=torch.float32, device=p.device), step=0)
            assert st['m'].numel() == n
            st['step'] += 1
            arr[k].p, arr[k].g, arr[k].m, arr[k].v = ptr(p), ptr(g), ptr(st['m']), ptr(st['v'])
            arr[k].n, arr[k].lr, arr[k].step = n, lr, st['step']
            arr[k].row_index, arr[k].row_len = ptr(rows), (p.shape[1] if rows is not None else 1)
            arr[k].zero_grad = int(bool(zero_grad))
            arr[k].p_f16 = int(p.dtype == torch.float16)
            keep.append((p, g, rows))
        self.eng.lib.check(self.eng.lib.dll.lk_adam_step(arr, len(segs), C.c_float(self.beta1), C.c_float(self.beta2),
                                                         C.c_float(self.eps), self.eng.stream), 'lk_adam_step')


def rays_from_pose(eng, cam7, pix_i, pix_j, intr, rays_o, rays_d):
    fx, fy, cx, cy = intr
    eng.lib.check(eng.lib.dll.lk_rays_from_pose(ptr(cam7), ptr(pix_i), ptr(pix_j), pix_i.shape[0], C.c_float(fx), C.c_float(fy),
                                                C.c_float(cx), C.c_float(cy), ptr(rays_o), ptr(rays_d), eng.stream),
                  'lk_rays_from_pose')


def pose_bwd(eng, cam7, pix_i, pix_j, intr, g_rays_o, g_rays_d, g_cam7):
    fx, fy, cx, cy = intr
    eng.lib.check(eng.lib.dll.lk_pose_bwd(ptr(cam7), ptr(pix_i), ptr(pix_j), pix_i.shape[0], C.c_float(fx), C.c_float(fy),
                                          C.c_float(cx), C.c_float(cy), ptr(g_rays_o), ptr(g_rays_d), ptr(g_cam7),
                                          eng.stream), 'lk_pose_bwd')


def inside_mask(eng, depth, mask, thr, scratch, depth_filtered=None):
    eng.lib.check(eng.lib.dll.lk_inside_mask(ptr(depth), depth.shape[0], ptr(mask), ptr(depth_filtered), ptr(thr), ptr(scratch),
                                             eng.stream), 'lk_inside_mask')


def compact(eng, mask, out_index, out_count):
    eng.lib.check(eng.lib.dll.lk_compact(ptr(mask), mask.shape[0], ptr(out_index), ptr(out_count), eng.stream), 'lk_compact')


def gather_rays(eng, depth_stack, color_stack, c2w_stack, frame_id, rnd, H, W, window, intr, out, r2_map_stack=None):
    """out: dict(rays_o, rays_d, gt_depth, gt_color[, pix_i, pix_j, r2_ray]); window = (H0, H1, W0, W1)."""
    H0, H1, W0, W1 = window
    fx, fy, cx, cy = intr
    R = rnd.shape[0]
    stride = c2w_stack.shape[-2] * 4
    eng.lib.check(eng.lib.dll.lk_gather_rays(ptr(depth_stack), ptr(color_stack), ptr(c2w_stack), stride, ptr(r2_map_stack),
                                             ptr(frame_id), ptr(rnd), R, H, W, H0, W0, W1 - W0,
                                             C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                             ptr(out['rays_o']), ptr(out['rays_d']), ptr(out['gt_depth']), ptr(out['gt_color']),
                                             ptr(out.get('pix_i')), ptr(out.get('pix_j')), ptr(out.get('r2_ray')), eng.stream),
                  'lk_gather_rays')


def frustum_rows(eng, pos, c2w, depth, intr, H, W, edge, return_mask=False, pending=False):
    """Mapper.get_mask_from_c2w on the device (lk_frustum_rows): int32 tensor of the selected row indices, ascending
    (and, with return_mask, the uint8 [N] membership flags).  One host sync (the count) - once per mapped frame; a second one if
    c2w lives on the device (the inverse is taken on the host, as the reference does: pass the host copy of the pose when there is one)."""
    import numpy as np
    fx, fy, cx, cy = intr
    N = pos.shape[0]
    c2w_h = c2w if isinstance(c2w, np.ndarray) else c2w.detach().cpu().numpy()
    w2c = np.linalg.inv(c2w_h.astype(np.float32)).astype(np.float32)     # float32 like the reference
    w12 = (C.c_float * 12)(*[float(x) for x in w2c[:3, :4].reshape(-1)])
    out = eng.empty(max(N, 1), dtype=torch.int32)
    cnt = eng.empty(1, dtype=torch.int32)
    sd = eng.empty(max(N, 1))
    sm = eng.empty(max(N, 1), dtype=torch.uint8)
    smax = eng.empty(1, dtype=torch.int32)
    eng.lib.check(eng.lib.dll.lk_frustum_rows(ptr(pos), N, w12, ptr(depth), H, W, C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                              C.c_float(cy), int(edge), ptr(sd), ptr(sm), ptr(smax), ptr(out), ptr(cnt),
                                              eng.stream), 'lk_frustum_rows')
    if pending:
        return _PendingRows(eng, out, cnt, sm[:N], return_mask)
    rows = out[:int(cnt.item())]
    return (rows, sm[:N]) if return_mask else rows


class _PendingRows:
    """frustum_rows(..., pending=True): the selection is enqueued and its count is on its way to pinned host memory; finish() waits for
    THAT copy only - whatever the caller enqueues in between (MapOptimizer.prepare: table fills, the call's batch assembly) runs on the
    device while the host waits for the count and builds its descriptors."""

    def __init__(self, eng, out, cnt, mask, return_mask):
        self.out, self.mask, self.return_mask = out, mask, return_mask
        if cnt.is_cuda:
            self.host = torch.empty(1, dtype=torch.int32).pin_memory()
            self.host.copy_(cnt, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(eng.device))
        else:
            self.host, self.event = cnt, None

    def finish(self):
        if self.event is not None:
            self.event.synchronize()
        rows = self.out[:int(self.host[0])]
        return (rows, self.mask) if self.return_mask else rows


def add_points(eng, knn, rays_o, rays_d, gt_depth, r2, near_surface, far_surface, n_add=3):
    """Geometry part of NeuralPointCloud.add_neural_points (lk_add_points): (accepted ray indices int32 [k], new points
    [k*n_add, 3]).  knn may be None / empty (first frame).  One host sync (the count)."""
    n = gt_depth.shape[0]
    per = r2.contiguous() if torch.is_tensor(r2) else None
    mask = eng.empty(max(n, 1), dtype=torch.uint8)
    idx = eng.empty(max(n, 1), dtype=torch.int32)
    cnt = eng.empty(1, dtype=torch.int32)
    pts = eng.empty(max(n, 1) * n_add, 3)
    h = knn.h if knn is not None else None
    eng.lib.check(eng.lib.dll.lk_add_points(h, ptr(rays_o.contiguous()), ptr(rays_d.contiguous()), ptr(gt_depth.contiguous()), n,
                                            C.c_float(0.0 if per is not None else float(r2)), ptr(per), C.c_float(near_surface),
                                            C.c_float(far_surface), n_add, ptr(mask), ptr(idx), ptr(cnt), ptr(pts), eng.stream),
                  'lk_add_points')
    k = int(cnt.item())
    return idx[:k], pts[:k * n_add]


def radius_maps(eng, color, thr, radius_add_max, radius_add_min, ratio):
    """Colour-gradient magnitude and the dynamic radius maps of one frame (lk_radius_maps):
    (grad_mag [H,W], r2_add [H,W], r2_query [H,W]) float32, radii SQUARED."""
    H, W = color.shape[:2]
    g, ra, rq = eng.empty(H, W), eng.empty(H, W), eng.empty(H, W)
    eng.lib.check(eng.lib.dll.lk_radius_maps(ptr(color.contiguous()), H, W, C.c_double(thr), C.c_double(radius_add_max),
                                             C.c_double(radius_add_min), C.c_double(ratio), ptr(g), ptr(ra), ptr(rq), eng.stream),
                  'lk_radius_maps')
    return g, ra, rq


def top_grad_pixels(eng, grad_mag, k, window, depth=None, depth_limit=False):
    """Pool of high-gradient pixels (lk_top_grad_pixels): ascending flat indices int32.  One host sync (the count)."""
    H, W = grad_mag.shape
    H0, H1, W0, W1 = window
    k = min(int(k), H * W)
    out = eng.empty(max(k, 1), dtype=torch.int32)
    cnt = eng.empty(1, dtype=torch.int32)
    eng.lib.check(eng.lib.dll.lk_top_grad_pixels(ptr(grad_mag), H, W, k, H0, H1, W0, W1, ptr(depth), int(bool(depth_limit)),
                                                 ptr(out), ptr(cnt), eng.stream), 'lk_top_grad_pixels')
    return out[:int(cnt.item())]
