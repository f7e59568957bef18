"""The benchmark workload: one "frame-equivalent" of the reference's per-frame work budget on a
synthetic 640x480 RGB-D room.

Replica room0 budget (configs/Replica/replica.yaml:21-22,25,33,36-37; configs/point_slam.yaml:92):
  tracking  40 iterations x 1500 rays   every frame
  mapping   300 iterations x 5000 rays  every 5th frame  -> 60 iterations per frame,
            geo_iter_ratio 0.4 -> 24 'geometry' + 36 'color' iterations
  => 360 000 rays per frame, S = 5 samples per ray, rel-pos colour MLP on, static radius 0.08.

The other two single-GPU budgets of BASELINE.json (Budget.tum / Budget.scannet; bench.py reports them under `workloads`):
  TUM_RGBD  (configs/TUM_RGBD/tum.yaml:4,7-9,16)       tracking 200 x 5000 rays per frame from the pool of the highest colour-gradient pixels,
            one leaf pose tensor; mapping 300 x 10 000 rays every 2nd frame -> 150 per frame (60 'geometry' + 90 'color') over a window
            of 10 keyframes; per-pixel dynamic query radius; plain colour model
  ScanNet   (configs/ScanNet/scannet.yaml:3-4,6-10,16,18) tracking 100 x 5000 at lr 5e-4; mapping 300 x 10 000 every 5th frame -> 60 per
            frame (geo_iter_ratio 0.3: 18 + 42) over 20 keyframes; surface ratios 0.96 / 1.04; exposure encoding (the 8 -> 128 -> 12 MLP,
            per-sample affine in the tracker, per-keyframe affine on the rendered logits in the mapper)
"""
from dataclasses import dataclass, field

import torch

from . import core, optim, parallel, steps, synthetic as syn
from .common import get_tensor_from_camera


MAP_LRS = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}     # mapping.stage.* (configs/point_slam.yaml:64-66; every config)


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
    # what makes the TUM / ScanNet configs different (all off = the Replica model)
    separate_lr: bool = True              # tracking.separate_LR
    dynamic_radius: bool = False          # use_dynamic_radius: per-pixel query radius from the colour gradient (lk_radius_maps per frame)
    grad_pool: bool = False               # tracking.sample_with_color_grad: tracking pixels from the 15 n highest-gradient pixels (lk_top_grad_pixels)
    exposure: bool = False                # model.encode_exposure
    near_surface: float = 0.98            # rendering.near_end_surface / far_end_surface
    far_surface: float = 1.02
    map_lrs: dict = field(default_factory=lambda: dict(MAP_LRS))

    @classmethod
    def tum(cls, **kw):
        return cls(name='tum_freiburg1_desk', track_iters=200, track_rays=5000, map_iters=150, map_geo_iters=60, map_rays=10000, window=10,
                   ignore_edge=20, cam_lr=0.002, rel_pos=False, every_frame=2, pixels_adding=5000, separate_lr=False, dynamic_radius=True,
                   grad_pool=True, **kw)

    @classmethod
    def scannet(cls, **kw):
        return cls(name='scannet_scene0000', track_iters=100, track_rays=5000, map_iters=60, map_geo_iters=18, map_rays=10000, window=20,
                   ignore_edge=20, cam_lr=0.0005, rel_pos=False, every_frame=5, pixels_adding=6000, separate_lr=False, dynamic_radius=True,
                   grad_pool=True, exposure=True, near_surface=0.96, far_surface=1.04, **kw)

    @property
    def rays_per_frame(self):
        return self.track_iters * self.track_rays + self.map_iters * self.map_rays


# pointcloud.* of configs/point_slam.yaml:128-131 (dynamic radii)
RADIUS_CFG = dict(color_grad_threshold=0.15, radius_add_max=0.08, radius_add_min=0.02, radius_query_ratio=2)


class FrameWorkload:
    """Device-resident scene + the two optimisers; step() = track one frame + its share of mapping."""

    def __init__(self, eng, budget=None, seed=1219, dist=None, cloud=None, intr=None):
        """cloud: (pos, geo, col, n_rooms) device tensors of an already built map (bench.py: the three budgets share one cloud).
        intr: camera dict as synthetic.TUM_INTR (the default: 640 x 480; the host-emulator test uses a few hundred pixels)."""
        self.dist = dist            # parallel.DistContext or None
        self.eng, self.b = eng, budget or Budget()
        b = self.b
        dev = eng.device
        self.cam = dict(intr or syn.TUM_INTR)
        self.intr = (self.cam['fx'], self.cam['fy'], self.cam['cx'], self.cam['cy'])
        self.H, self.W = self.cam['H'], self.cam['W']
        self.cfg = core.RenderCfg(rel_pos=b.rel_pos, exposure=b.exposure, near_surface=b.near_surface, far_surface=b.far_surface)
        W0 = syn.default_weights(seed, rel_pos=b.rel_pos, exposure=b.exposure)
        self.dec = core.DecoderBlob(eng).pack(W0)
        if cloud is not None:
            pos, geo, col, self.n_rooms = cloud
        elif b.online_cloud:
            pos, geo, col, self.n_rooms = syn.build_cloud_online(eng, b.n_points, seed=seed, intr=self.cam)
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
            d, c, p = syn.render_frame(3 * k, intr=self.cam, device=dev, holes=0.02, seed=seed)
            ds.append(d); cs.append(c); ps.append(syn.pose_in_room(p, room))
        self.depth_stack = torch.stack(ds).contiguous()
        self.color_stack = torch.stack(cs).contiguous()
        self.c2w_stack = torch.stack(ps).contiguous()
        self.c2w_host = [p.detach().cpu().numpy() for p in ps]      # the mapped frame's pose is known to the host (frustum selection)
        # dynamic radii: the squared query-radius map of every keyframe of the window (Mapper.py:854-872 keeps them with the keyframes);
        # the tracked frame's maps are recomputed per step (Tracker.py:243-258), as the gradient-pixel pool is
        self.r2_stack = None
        if b.dynamic_radius:
            self.r2_stack = torch.stack([self._radius_maps(k)[2] for k in range(b.window)]).contiguous()
        self.frames = (self.depth_stack, self.color_stack, self.c2w_stack, self.r2_stack)
        # exposure encoding: mlp_exposure as the torch module that owns its parameters (the kernels step them in place), one feature
        # per keyframe of the window (Mapper.py:588-607) - the mapped frame's is the trainable one - and the tracker's copy of the shared one
        self.mlp_exposure, self.exposure_feats = None, None
        if b.exposure:
            m = torch.nn.Sequential(torch.nn.Linear(8, 128), torch.nn.Softplus(beta=100), torch.nn.Linear(128, 12)).to(dev)
            with torch.no_grad():
                for lin, n in ((m[0], 'linear1'), (m[2], 'linear2')):
                    lin.weight.copy_(W0[f'color_decoder.mlp_exposure.{n}.weight']); lin.bias.copy_(W0[f'color_decoder.mlp_exposure.{n}.bias'])
            self.mlp_exposure = m
            g = torch.Generator().manual_seed(seed + 5)
            self.exposure_feats = [(0.1 * torch.randn(8, generator=g)).to(dev).requires_grad_(True) for _ in range(b.window)]
        # rows optimised by the mapper = frustum selection of the frame being mapped (Mapper.py:165-217, 498-512),
        # recomputed at every step like the reference does at every optimize_map call
        self.rows = optim.frustum_rows(eng, self.pos[:self.n], self.c2w_stack[0], self.depth_stack[0], self.intr, self.H, self.W, b.frustum_edge)
        self.mapper = steps.MapOptimizer(eng, self.cfg, self.dec, self.knn, self.pos, self.geo, self.col, self.rows,
                                         b.map_rays, b.map_lrs, w_color=0.1, dist=dist, dynamic_radius=b.dynamic_radius,
                                         exposure=(self.mlp_exposure, self.exposure_feats) if b.exposure else None)
        self.tracker = steps.TrackOptimizer(eng, self.cfg, self.dec, self.knn, self.pos, self.geo, self.col,
                                            b.track_rays, b.cam_lr, separate_lr=b.separate_lr, w_color=0.5, dist=dist,
                                            dynamic_radius=b.dynamic_radius)
        # pixel draws happen on the device, as the reference's do (select_uv: torch.randint(..., device=device), common.py:156-172):
        # 300 000 host-side draws + their upload cost 0.9 ms of a 22 ms step with the GPU idle
        # (multi-GPU: the mapping draws differ per rank - every rank renders its own rays of the shared iteration - the tracking draws
        # are the same everywhere: tracking is replicated, steps.TrackOptimizer)
        self.gen = torch.Generator(device=dev).manual_seed(seed + (dist.rank if dist is not None else 0))
        self.gen_track = torch.Generator(device=dev).manual_seed(seed + 7919)          # shared by the ranks: tracking draws, insertion
        # pixels // window rays of every keyframe per iteration (Mapper.py:417-418); with exposure encoding grouped by keyframe as
        # slam.Mapper.optimize_map lays them out (the loss kernel sums d affine over the lanes of a keyframe)
        self._fid = ((torch.arange(b.map_rays, dtype=torch.int32) // max(1, b.map_rays // b.window)).clamp(max=b.window - 1) if b.exposure
                     else (torch.arange(b.map_rays, dtype=torch.int32) % b.window)).to(dev)
        self.cam0 = get_tensor_from_camera(self.c2w_stack[0]).to(dev)
        self.map_log = eng.zeros(b.map_iters, 4)
        self.frame_no = 0
        self.img_state = None               # RenderState of the full-frame render (full step)
        self.n_added = 0
        # The full-frame render of a mapped frame (Mapper.py:966-969: an image for the run's output folder, nothing of the loop consumes it) runs
        # on a stream of its own, beside the NEXT FRAME'S TRACKING - forward-only work over the map, and tracking only reads the map too.
        # Whatever writes the map or the decoders next - the next step's mapping iterations (features, decoder blob, fragments), a mapped
        # frame's insertion - waits for it first (step(): _join_render).
        self.render_stream = torch.cuda.Stream(eng.device) if eng.device.type == 'cuda' else None

    def _radius_maps(self, k):
        """(grad_mag, r2_add, r2_query) of keyframe k (lk_radius_maps)."""
        return optim.radius_maps(self.eng, self.color_stack[k], RADIUS_CFG['color_grad_threshold'], RADIUS_CFG['radius_add_max'],
                                 RADIUS_CFG['radius_add_min'], RADIUS_CFG['radius_query_ratio'])

    def _join_render(self):
        """The launch stream waits for the last mapped frame's full-frame render: called before anything writes what the render reads."""
        if self.render_stream is not None:
            torch.cuda.current_stream(self.eng.device).wait_stream(self.render_stream)

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
        ro, rd = syn.pixel_rays(self.c2w_stack[k], i, j, self.cam)
        gd = self.depth_stack[k].reshape(-1)[px]
        _, pts = optim.add_points(eng, self.knn, ro, rd, gd, float(torch.tensor(0.04 ** 2, dtype=torch.float32)), b.near_surface, b.far_surface, 3)
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
        # several ranks: each renders ONE contiguous range of whole 3000-ray groups (parallel.ray_range) and the ranges are
        # exchanged with one sum all-reduce of the zero-padded outputs - the frame of slam.Renderer.render_img on every rank
        dist = self.dist if (self.dist is not None and self.dist.world > 1) else None
        lo, hi = parallel.ray_range(H * W, dist.rank, dist.world, 3000) if dist is not None else (0, H * W)
        if self.img_state is None:
            jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=eng.device), torch.arange(W, dtype=torch.float32, device=eng.device), indexing='ij')
            self._img_ij = (ii.reshape(-1)[lo:hi].contiguous(), jj.reshape(-1)[lo:hi].contiguous())
            self.img_state = core.RenderState(eng, hi - lo, self.cfg.S)
            self.img_full = [eng.zeros(H * W), eng.zeros(H * W), eng.zeros(H * W, 3)] if dist is not None else None
        ro, rd = syn.pixel_rays(self.c2w_stack[k], *self._img_ij, self.cam)
        gd = self.depth_stack[k].reshape(-1)[lo:hi].contiguous()
        r2 = self.r2_stack[k].reshape(-1)[lo:hi].contiguous() if self.r2_stack is not None else None          # (dynamic radii: the frame's own query-radius map)
        core.render_forward(eng, self.cfg, self.img_state, ro, rd, gd, self.knn, self.pos, self.geo, self.col, self.dec, 'color', stats_chunk=3000,
                            r2_ray=r2)
        if dist is not None:
            st = self.img_state
            for t in self.img_full:
                t.zero_()
            self.img_full[0][lo:hi], self.img_full[1][lo:hi], self.img_full[2][lo:hi] = st.depth, st.var, st.color
            self.img_frame = dist.gather_ranges(self.img_full, lo, hi)
        return self.img_state

    def step(self, full=False):
        """One frame-equivalent of the budget (Replica: 40 tracking iterations + 60 mapping iterations, 24 geometry + 36 colour).  full: every
        `every_frame`-th step is a MAPPED frame and also runs mapped_frame_extras / render_frame - the work the reference does
        once per mapped frame beside its 300 iterations (their 60-per-frame share is in every step)."""
        b, eng = self.b, self.eng
        H, W = self.H, self.W
        e = min(b.ignore_edge, H // 4)
        win = (e, H - e, e, W - e)
        k = self.frame_no % b.window
        r2_map = None
        win_t = win
        if b.dynamic_radius or b.grad_pool:       # the tracked frame's image pre-pass (Tracker.py:243-268), every frame
            grad, _, r2q = self._radius_maps(k)
            if b.dynamic_radius:
                r2_map = r2q
                self.r2_stack[k].copy_(r2q)          # (the mapper's copy of this keyframe's map: same values, refreshed like an online run's)
        if b.grad_pool:
            # n of the 15 n highest-gradient pixels of the window per iteration, without replacement (common.py:198-234, Tracker.py:126-139):
            # the n largest of a row of uniform draws, drawn and selected on the device (slam.Tracker.track_frame)
            pool = optim.top_grad_pixels(eng, grad, 15 * b.track_rays, win, self.depth_stack[k], False)
            # (the pool is the 15 n largest gradients of the WHOLE image restricted to the window and to pixels with a depth - common.py:198-234 -
            # so it can be smaller than 15 n; the reference then draws min(n, pool) pixels, this workload's batches have a fixed size)
            assert int(pool.numel()) >= b.track_rays, f'gradient-pixel pool of {int(pool.numel())} pixels < {b.track_rays} tracking rays on this frame'
            u = torch.rand(b.track_iters, pool.numel(), generator=self.gen_track, device=eng.device)
            rnd_t = pool[u.topk(b.track_rays, dim=1).indices].contiguous()
            win_t = (0, H, 0, W)
        else:
            rnd_t = self._draws(b.track_iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]), self.gen_track)
        xt = None
        if b.exposure:                              # this frame's exposure feature starts from the keyframe's (Tracker.py:280-283)
            self._xfeat_track = self.exposure_feats[k].detach().clone().requires_grad_(True)
            xt = (self.mlp_exposure, self._xfeat_track)
        best, tlog = self.tracker.track(self.cam0, self.depth_stack[k], self.color_stack[k], b.track_iters, win_t, self.intr, rnd_t,
                                        r2_map=r2_map, exposure=xt)
        # the mapping iterations below step feature rows, the decoder blob and its fragments: the full-frame render of the last mapped
        # frame (its own stream) has overlapped this frame's tracking and must be through before they start
        self._join_render()
        mapped = full and self.frame_no % b.every_frame == 0
        if mapped:
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
        # end of the optimize_map call: the stepped exposure feature goes back to its keyframe's tensor (Mapper.py:772-777 for the rows, the
        # feature with them) - the next frame's tracker starts from it; round 5 left it in the optimiser's stacked copy (advisor)
        self.mapper.finish()
        if mapped:
            if self.render_stream is not None:
                self.render_stream.wait_stream(torch.cuda.current_stream(eng.device))
                with torch.cuda.stream(self.render_stream):
                    self.render_frame(k)
            else:
                self.render_frame(k)
        self.frame_no += 1
        return best, tlog, self.map_log
