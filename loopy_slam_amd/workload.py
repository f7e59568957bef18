"""The benchmark workload: one "frame-equivalent" of the reference's per-frame work budget on a
synthetic 640x480 RGB-D room.

Replica room0 budget (configs/Replica/replica.yaml:21-22,25,33,36-37; configs/point_slam.yaml:92):
  tracking  40 iterations x 1500 rays   every frame
  mapping   300 iterations x 5000 rays  every 5th frame  -> 60 iterations per frame,
            geo_iter_ratio 0.4 -> 24 'geometry' + 36 'color' iterations
  => 360 000 rays per frame, S = 5 samples per ray, rel-pos colour MLP on, static radius 0.08.
"""
import os
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
    online_cloud: bool = True             # the map as an online run builds it (radius-de-duplicated insertion, one room per 100 k points:
                                          # synthetic.build_cloud_online); False: random pixels of 24 views packed into one room
    every_frame: int = 5                  # mapping.every_frame: the full step inserts points / re-indexes / renders the frame every 5th frame
    pixels_adding: int = 6000             # mapping.pixels_adding

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
        if b.online_cloud:
            pos, geo, col, self.n_rooms = syn.build_cloud_online(eng, b.n_points, seed=seed)
        else:
            pos, geo, col = (t.to(dev) for t in syn.build_cloud(b.n_points, device='cpu', seed=seed))
            self.n_rooms = 1
        # tables at a fixed capacity (the full step keeps inserting points, NeuralPointCloud-style); rows >= n are never indexed
        self.n = int(pos.shape[0])
        b.n_points = self.n
        self.capacity = self.n + 40 * 3 * b.pixels_adding
        self.pos, self.geo, self.col = eng.zeros(self.capacity, 3), eng.zeros(self.capacity, 32), eng.zeros(self.capacity, 32)
        self.pos[:self.n], self.geo[:self.n], self.col[:self.n] = pos, geo, col
        del pos, geo, col
        self.knn = core.KnnIndex(eng, capacity=self.capacity)
        self.knn.build(self.pos[:self.n])
        # keyframe window: stacked depth / colour / pose; the camera works in the LAST room of a multi-room map (the part of the map
        # it does not see is dead weight for the index and the tables, as in a long sequence)
        room = self.n_rooms - 1
        ds, cs, ps = [], [], []
        for k in range(b.window):
            d, c, p = syn.render_frame(3 * k, device=dev, holes=0.02, seed=seed)
            ds.append(d); cs.append(c); ps.append(syn.pose_in_room(p, room))
        self.depth_stack = torch.stack(ds).contiguous()
        self.color_stack = torch.stack(cs).contiguous()
        self.c2w_stack = torch.stack(ps).contiguous()
        self.c2w_host = [p.detach().cpu().numpy() for p in ps]      # the mapped frame's pose is known to the host (frustum selection)
        self.frames = (self.depth_stack, self.color_stack, self.c2w_stack, None)
        # rows optimised by the mapper = frustum selection of the frame being mapped (Mapper.py:165-217, 498-512),
        # recomputed at every step like the reference does at every optimize_map call
        self.rows = optim.frustum_rows(eng, self.pos[:self.n], self.c2w_stack[0], self.depth_stack[0], self.intr, self.H, self.W, b.frustum_edge)
        self.mapper = steps.MapOptimizer(eng, self.cfg, self.dec, self.knn, self.pos, self.geo, self.col, self.rows,
                                         b.map_rays, MAP_LRS, w_color=0.1, dist=dist)
        self.tracker = steps.TrackOptimizer(eng, self.cfg, self.dec, self.knn, self.pos, self.geo, self.col,
                                            b.track_rays, b.cam_lr, separate_lr=True, w_color=0.5, dist=dist)
        # pixel draws happen on the device, as the reference's do (select_uv: torch.randint(..., device=device), common.py:156-172):
        # 300 000 host-side draws + their upload cost 0.9 ms of a 22 ms step with the GPU idle
        # (multi-GPU: the mapping draws differ per rank - every rank renders its own rays of the shared iteration - the tracking draws
        # are the same everywhere: tracking is replicated, steps.TrackOptimizer)
        self.gen = torch.Generator(device=dev).manual_seed(seed + (dist.rank if dist is not None else 0))
        self.gen_track = torch.Generator(device=dev).manual_seed(seed + 7919)          # shared by the ranks: tracking draws, insertion
        self._fid = (torch.arange(b.map_rays, dtype=torch.int32) % b.window).to(dev)      # pixels // window frames each
        self.cam0 = get_tensor_from_camera(self.c2w_stack[0]).to(dev)
        self.map_log = eng.zeros(b.map_iters, 4)
        self.frame_no = 0
        self.img_state = None               # RenderState of the full-frame render (full step)
        self.n_added = 0
        # The full-frame render of a mapped frame (Mapper.py:966-969: an image for the run's output folder, nothing of the loop consumes it) runs
        # on a stream of its own, beside the next frames' tracking - forward-only work over a map that nobody changes until the next mapped
        # frame's insertion, which waits for it.  LOOPY_RENDER_INLINE=1: on the launch stream, behind the mapping iterations (A/B).
        self.render_stream = None
        if eng.device.type == 'cuda' and os.environ.get('LOOPY_RENDER_INLINE') != '1':
            self.render_stream = torch.cuda.Stream(eng.device)

    def _draws(self, iters, R, n, gen=None):
        return torch.randint(0, n, (iters, R), generator=gen or self.gen, dtype=torch.int32, device=self.eng.device)

    def mapped_frame_extras(self, k):
        """What a MAPPED frame does around its iterations (every `every_frame`-th frame): point insertion of `pixels_adding`
        random pixels with the radius test against the current map + feature rows for the new points + index rebuild
        (Mapper.py:421-482, neural_point.py:1557-1631), and the full-frame render after the optimisation
        (Renderer.render_img, Mapper.py:966-969)."""
        b, eng = self.b, self.eng
        H, W = self.H, self.W
        # (multi-GPU: insertion is REPLICATED work - the same pixels and feature draws on every rank keep the map replicas identical)
        px = torch.randint(0, H * W, (b.pixels_adding,), generator=self.gen_track, device=eng.device)
        i, j = (px % W).float(), torch.div(px, W, rounding_mode='floor').float()
        ro, rd = syn.pixel_rays(self.c2w_stack[k], i, j)
        gd = self.depth_stack[k].reshape(-1)[px]
        _, pts = optim.add_points(eng, self.knn, ro, rd, gd, float(torch.tensor(0.04 ** 2, dtype=torch.float32)), 0.98, 1.02, 3)
        m = min(int(pts.shape[0]), self.capacity - self.n)
        if m:
            self.pos[self.n:self.n + m] = pts[:m]
            self.geo[self.n:self.n + m] = 0.1 * torch.randn(m, 32, generator=self.gen_track, device=eng.device)
            self.col[self.n:self.n + m] = 0.1 * torch.randn(m, 32, generator=self.gen_track, device=eng.device)
            self.n += m
            self.n_added += m
            self.knn.build(self.pos[:self.n])

    def render_frame(self, k):
        """Renderer.render_img of keyframe k: all H*W rays in one fused pass."""
        eng, H, W = self.eng, self.H, self.W
        if self.img_state is None:
            jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=eng.device), torch.arange(W, dtype=torch.float32, device=eng.device), indexing='ij')
            self._img_ij = (ii.reshape(-1).contiguous(), jj.reshape(-1).contiguous())
            self.img_state = core.RenderState(eng, H * W, self.cfg.S)
        ro, rd = syn.pixel_rays(self.c2w_stack[k], *self._img_ij)
        gd = self.depth_stack[k].reshape(-1).contiguous()
        core.render_forward(eng, self.cfg, self.img_state, ro, rd, gd, self.knn, self.pos, self.geo, self.col, self.dec, 'color', stats_chunk=3000)
        return self.img_state

    def step(self, full=False):
        """One frame-equivalent: 40 tracking iterations + 60 mapping iterations (24 geometry + 36 colour).  full: every
        `every_frame`-th step is a MAPPED frame and also runs mapped_frame_extras / render_frame - the work the reference does
        once per mapped frame beside its 300 iterations (their 60-per-frame share is in every step)."""
        b, eng = self.b, self.eng
        H, W = self.H, self.W
        e = min(b.ignore_edge, H // 4)
        win = (e, H - e, e, W - e)
        rnd_t = self._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]), self.gen_track)
        k = self.frame_no % b.window
        best, tlog = self.tracker.track(self.cam0, self.depth_stack[k], self.color_stack[k], b.track_iters, win, self.intr, rnd_t)
        mapped = full and self.frame_no % b.every_frame == 0
        if mapped:
            if self.render_stream is not None:      # the insertion changes the map and its index: the last frame's render must be through
                torch.cuda.current_stream(eng.device).wait_stream(self.render_stream)
            self.mapped_frame_extras(k)
        rnd_m = self._draws(b.map_iters, b.map_rays, H * W)
        fid = self._fid
        # the gradient-table fills and the batch assembly of the mapping call do not depend on the selected rows: enqueued before the
        # selection's count read-back, the device works through them while the host waits and builds the call (MapOptimizer.prepare)
        # (order on the stream: selection, count on its way to the host, THEN fills + assembly - the host waits for the count alone)
        sel = optim.frustum_rows(eng, self.pos[:self.n], self.c2w_host[k], self.depth_stack[k], self.intr, H, W, b.frustum_edge,
                                 return_mask=True, pending=True)
        prepared = self.mapper.prepare(b.map_iters, b.map_geo_iters, self.frames, rnd_m, fid, (0, H, 0, W), self.intr, H, W, self.map_log)
        self.rows, row_mask = sel.finish()
        self.mapper.new_frame(self.rows, row_mask, zero=not prepared)
        self.mapper.run(b.map_iters, b.map_geo_iters, self.frames, rnd_m, fid, (0, H, 0, W), self.intr, H, W, self.map_log)
        if mapped:
            if self.render_stream is not None:
                self.render_stream.wait_stream(torch.cuda.current_stream(eng.device))
                with torch.cuda.stream(self.render_stream):
                    self.render_frame(k)
            else:
                self.render_frame(k)
        self.frame_no += 1
        return best, tlog, self.map_log
