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


def loss_tracker(eng, st, gt_depth, gt_color, w_color, use_color, d_depth, d_color, out4, scratch):
    """Tracker.py:169-191 (handle_dynamic).  scratch: R+8 floats."""
    R = gt_depth.shape[0]
    eng.lib.check(eng.lib.dll.lk_loss_tracker(R, ptr(st.depth), ptr(st.var), ptr(st.color), ptr(gt_depth), ptr(gt_color),
                                              C.c_float(w_color), int(bool(use_color)), ptr(d_depth), ptr(d_color),
                                              ptr(out4), ptr(scratch), eng.stream), 'lk_loss_tracker')


class Adam:
    """torch.optim.Adam (amsgrad=False, weight_decay=0) over flat fp32 tensors, one launch per step.
    Each segment is (param, grad) with its own lr and step count; a segment whose lr group has no
    gradient this step is simply not passed (torch skips parameters whose .grad is None)."""

    def __init__(self, eng, beta1=0.9, beta2=0.999, eps=1e-8):
        self.eng, self.beta1, self.beta2, self.eps = eng, beta1, beta2, eps
        self.state = {}

    def step(self, segs):
        """segs: list of (key, param, grad, lr).  param/grad: contiguous fp32 tensors of equal numel."""
        assert len(segs) <= _ffi.ADAM_MAX_SEG
        arr = (AdamSeg * max(1, len(segs)))()
        keep = []
        for k, (key, p, g, lr) in enumerate(segs):
            st = self.state.get(key)
            if st is None:
                st = self.state[key] = dict(m=torch.zeros_like(p), v=torch.zeros_like(p), step=0)
            st['step'] += 1
            arr[k].p, arr[k].g, arr[k].m, arr[k].v = ptr(p), ptr(g), ptr(st['m']), ptr(st['v'])
            arr[k].n, arr[k].lr, arr[k].step = p.numel(), lr, st['step']
            keep.append((p, g))
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


def inside_mask(eng, depth, mask, thr, scratch):
    eng.lib.check(eng.lib.dll.lk_inside_mask(ptr(depth), depth.shape[0], ptr(mask), ptr(thr), ptr(scratch), eng.stream),
                  'lk_inside_mask')


def compact(eng, mask, out_index, out_count):
    eng.lib.check(eng.lib.dll.lk_compact(ptr(mask), mask.shape[0], ptr(out_index), ptr(out_count), eng.stream), 'lk_compact')
