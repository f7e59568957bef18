"""The benchmark workload (loopy_slam_amd/workload.py: what bench.py times) at toy budgets: the three budgets of BASELINE.json - Replica
(rel-pos colour model, separate_LR), TUM (dynamic radii, gradient-pixel pool, one leaf pose) and ScanNet (exposure encoding, 0.96 / 1.04)
- step through both loops, the mapped-frame extras and the full-frame render, and leave finite losses that the iterations lower."""
import numpy as np
import pytest
import torch

from loopy_slam_amd import workload, synthetic as syn
from util import make_engine, backends


def _budget(name, backend):
    # (tracking rays of the gradient-pool budgets: the pool is the 15 n largest gradients of the whole frame INSIDE the window - on the
    # 640 x 480 synthetic room 22 396 of the 30 000 largest, none of the 720 largest: n = 2 000 there, 48 on the emulator's 64 x 48 frames)
    small = dict(track_iters=3 if backend == 'hip' else 2, track_rays=(48 if backend == 'hip' else 32) if (backend == 'emu' or name == 'replica') else 2000,
                 map_iters=5 if backend == 'hip' else 4, map_geo_iters=2, map_rays=120 if backend == 'hip' else 48,
                 n_points=6000 if backend == 'hip' else 4000, pixels_adding=200 if backend == 'hip' else 120)
    if name == 'replica':
        return workload.Budget(window=3, every_frame=2, **small)
    mk = workload.Budget.tum if name == 'tum' else workload.Budget.scannet
    b = mk()
    for k, v in small.items():
        setattr(b, k, v)
    b.window, b.every_frame = 3, 2
    return b


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('name', ('replica', 'tum', 'scannet'))
def test_frame_workload_steps(backend, name):
    eng = make_engine(backend)
    b = _budget(name, backend)
    # (the host emulator runs every lane as a fiber: a 640 x 480 frame's full render would take half an hour there)
    cam = dict(syn.TUM_INTR) if backend == 'hip' else dict(H=48, W=64, fx=51.7, fy=51.6, cx=31.9, cy=25.5)
    pos, geo, col = (t.to(eng.device) for t in syn.build_cloud(b.n_points, device='cpu', seed=3, intr=cam))
    b.ignore_edge = 20 if backend == 'hip' else 4
    wl = workload.FrameWorkload(eng, b, cloud=(pos, geo, col, 1), intr=cam)
    assert (wl.r2_stack is not None) == b.dynamic_radius and (wl.mlp_exposure is not None) == b.exposure
    geo0, blob0 = wl.geo[:wl.n].clone(), wl.dec.blob.clone()
    if b.exposure:          # before the steps: the window's exposure features and the module the kernels step in place
        xf0 = [f.detach().clone() for f in wl.exposure_feats]
        xw0 = [p.detach().clone() for p in wl.mlp_exposure.parameters()]
    logs = []
    for k in range(3 if backend == 'hip' else 2):          # (every_frame 2: steps 0 and 2 are mapped frames; the emulator runs two)
        best, tlog, mlog = wl.step(full=True)
        if eng.device.type == 'cuda':
            torch.cuda.synchronize()
        t, m = tlog.cpu().numpy(), mlog.cpu().numpy()
        assert np.isfinite(t).all() and np.isfinite(m).all() and np.isfinite(best.cpu().numpy()).all()
        assert (t[:, 3] > 0.4 * b.track_rays).all() and (m[:, 3] > 0.4 * b.map_rays).all()          # rays that took part in the losses
        assert (m[:b.map_geo_iters, 2] == 0).all() and (m[b.map_geo_iters:, 2] > 0).all()            # colour term only in the 'color' stage
        logs.append(m[:, 0].copy())
    assert wl.n_added > 0 and wl.n == b.n_points + wl.n_added                      # frames 0 and 2 were mapped frames: points inserted
    assert wl.img_state is not None and np.isfinite(wl.img_state.depth.cpu().numpy()).all()       # ... and rendered
    assert float((wl.geo[:b.n_points] - geo0).abs().max()) > 1e-3 and not torch.equal(wl.dec.blob, blob0)
    if b.exposure:
        # Mapper.py:524-570, 588-607: the mapped frame's own feature (the last of the window) and mlp_exposure are Adam parameters of the
        # mapping call, the other keyframes' features are constants - trained means MOVED, frozen means bit for bit where it was
        assert float((wl.exposure_feats[-1].detach() - xf0[-1]).abs().max()) > 1e-4
        assert all(torch.equal(f.detach(), f0) for f, f0 in zip(wl.exposure_feats[:-1], xf0[:-1]))
        assert all(float((p.detach() - p0).abs().max()) > 1e-5 for p, p0 in zip(wl.mlp_exposure.parameters(), xw0))
