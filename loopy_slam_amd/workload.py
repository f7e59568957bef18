"""The benchmark workload: one "frame-equivalent" of the reference's per-frame work budget on a
synthetic 640x480 RGB-D room.

Replica room0 budget (configs/Replica/replica.yaml:21-22,25,33,36-37; configs/point_slam.yaml:92):
  tracking  40 iterations x 1500 rays   every frame
  mapping   300 iterations x 5000 rays  every 5th frame  -> 60 iterations per frame,
            geo_iter_ratio 0.4 -> 24 'geometry' + 36 'color' iterations
  => 360 000 rays per frame, S = 5 samples per ray, rel-pos colour MLP on, static radius 0.08.
"""
from dataclasses import dataclass

import torch

from . import core, optim, steps, synthetic as syn
from .common import get_tensor_from_camera


@dataclass
class Budget:
    name: str = 'replica_room0'
    track_iters: int = 40
    track_rays: int = 1500
    map_iters: int = 60
    map_geo_iters: int = 24
    map_rays: int = 5000
    n_points: int = 100_000
    window: int = 12                      # mapping_window_size keyframes sampled per iteration
    ignore_edge: int = 100                # tracking.ignore_edge_W/H on Replica (scaled image: see below)
    cam_lr: float = 0.002
    rel_pos: bool = True
    frustum_edge: int = -4                # mapping.frustum_edge (configs/point_slam.yaml:67)

    @property
    def rays_per_frame(self):
        return self.track_iters * self.track_rays + self.map_iters * self.map_rays


MAP_LRS = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}     # mapping.stage.* (replica.yaml)


class FrameWorkload:
    """Device-resident scene + the two optimisers; step() = track one frame + its share of mapping."""

    def __init__(self, eng, budget=None, seed=1219, dist=None):
        self.eng, self.b = eng, budget or Budget()
        b = self.b
        dev = eng.device
        self.intr = (syn.TUM_INTR['fx'], syn.TUM_INTR['fy'], syn.TUM_INTR['cx'], syn.TUM_INTR['cy'])
        self.H, self.W = syn.TUM_INTR['H'], syn.TUM_INTR['W']
        self.cfg = core.RenderCfg(rel_pos=b.rel_pos)
        self.dec = core.DecoderBlob(eng).pack(syn.default_weights(seed, rel_pos=b.rel_pos))
        pos, geo, col = syn.build_cloud(b.n_points, device='cpu', seed=seed)
        self.pos, self.geo, self.col = pos.to(dev), geo.to(dev), col.to(dev)
        self.knn = core.KnnIndex(eng, capacity=b.n_points)
        self.knn.build(self.pos)
        # keyframe window: stacked depth / colour / pose
        ds, cs, ps = [], [], []
        for k in range(b.window):
            d, c, p = syn.render_frame(3 * k, device=dev, holes=0.02, seed=seed)
            ds.append(d); cs.append(c); ps.append(p)
        self.depth_stack = torch.stack(ds).contiguous()
        self.color_stack = torch.stack(cs).contiguous()
        self.c2w_stack = torch.stack(ps).contiguous()
        self.frames = (self.depth_stack, self.color_stack, self.c2w_stack, None)
        # rows optimised by the mapper = frustum selection of the frame being mapped (Mapper.py:165-217, 498-512),
        # recomputed at every step like the reference does at every optimize_map call
        self.rows = optim.frustum_rows(eng, self.pos, self.c2w_stack[0], self.depth_stack[0], self.intr, self.H, self.W, b.frustum_edge)
        self.mapper = steps.MapOptimizer(eng, self.cfg, self.dec, self.knn, self.pos, self.geo, self.col, self.rows,
                                         b.map_rays, MAP_LRS, w_color=0.1, dist=dist)
        self.tracker = steps.TrackOptimizer(eng, self.cfg, self.dec, self.knn, self.pos, self.geo, self.col,
                                            b.track_rays, b.cam_lr, separate_lr=True, w_color=0.5, dist=dist)
        # pixel draws happen on the device, as the reference's do (select_uv: torch.randint(..., device=device), common.py:156-172):
        # 300 000 host-side draws + their upload cost 0.9 ms of a 22 ms step with the GPU idle
        # (multi-GPU: the mapping draws differ per rank - every rank renders its own rays of the shared iteration - the tracking draws
        # are the same everywhere: tracking is replicated, steps.TrackOptimizer)
        self.gen = torch.Generator(device=dev).manual_seed(seed + (dist.rank if dist is not None else 0))
        self.gen_track = torch.Generator(device=dev).manual_seed(seed + 7919)
        self._fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(dev)      # pixels // window frames each
        self.cam0 = get_tensor_from_camera(self.c2w_stack[0]).to(dev)
        self.map_log = eng.zeros(b.map_iters, 4)
        self.frame_no = 0

    def _draws(self, iters, R, n, gen=None):
        return torch.randint(0, n, (iters, R), generator=gen or self.gen, dtype=torch.int32, device=self.eng.device)

    def step(self):
        """One frame-equivalent: 40 tracking iterations + 60 mapping iterations (24 geometry + 36 colour)."""
        b, eng = self.b, self.eng
        H, W = self.H, self.W
        e = min(b.ignore_edge, H // 4)
        win = (e, H - e, e, W - e)
        rnd_t = self._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]), self.gen_track)
        k = self.frame_no % b.window
        best, tlog = self.tracker.track(self.cam0, self.depth_stack[k], self.color_stack[k], b.track_iters, win, self.intr, rnd_t)
        rnd_m = self._draws(b.map_iters, b.map_rays, H * W)
        fid = self._fid
        self.rows, row_mask = optim.frustum_rows(eng, self.pos, self.c2w_stack[k], self.depth_stack[k], self.intr, H, W, b.frustum_edge,
                                                 return_mask=True)
        self.mapper.new_frame(self.rows, row_mask)
        self.mapper.run(b.map_iters, b.map_geo_iters, self.frames, rnd_m, fid, (0, H, 0, W), self.intr, H, W, self.map_log)
        self.frame_no += 1
        return best, tlog, self.map_log
