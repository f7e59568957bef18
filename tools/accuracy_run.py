#!/usr/bin/env python3
"""Accuracy of the drop-in pipeline against the CPU oracle's SLAM loop on the same synthetic RGB-D sequence (BASELINE config 1:
"50 frames, 500 rays/iter"), with the reference's own metrics:

  * ATE RMSE [cm]: translational RMSE of the camera centres after Horn alignment (src/tools/eval_ate.py:44-79, 195-234),
  * rendered-depth L1 [cm]: |sensor depth - re-rendered depth| at the ESTIMATED poses of every `every_frame`-th frame over the pixels
    with a reading (src/Mapper.py:1146-1182), here on a pixel grid of stride 4 for both pipelines (a full frame is ~15 s on the CPU oracle),
  * and what a tracker that does nothing would score on the sequence (tests/oracle_slam.py: prior_only_metrics).

    python tools/accuracy_run.py --pipeline oracle --config configs/Synthetic/room.yaml --frames 50 --rays 500 --out tests/golden/accuracy_room_oracle.json
    python tools/accuracy_run.py --pipeline hip    --config configs/Synthetic/room.yaml --frames 50 --rays 500 --out gpurun_out/accuracy_room_hip.json

`--pipeline oracle` is CPU only (it imports oracle/ and tests/oracle_slam.py: test infrastructure, run in the build container, its result
committed as a fixture); `--pipeline hip` is the product (loopy_slam_amd.slam.Point_SLAM on cuda:0); `--pipeline emu` the product on the
test-only host emulator.  Both pipelines start from the same random-init decoders (the pretrained middle_fine.pt is not available
offline) and read the same frames (synthetic.handheld_pose: 1-2 cm and 0.5-1.2 degrees per frame with a changing velocity)."""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def make_cfg(path, frames, rays, iters_scale=1.0, iters_first=None, seed=None, color_refine=None, scene='furnished'):
    from loopy_slam_amd import config
    cfg = copy.deepcopy(config.load_config(path, os.path.join(ROOT, 'configs/point_slam.yaml')))
    cfg['data']['n_frames'] = frames
    cfg['data']['motion'] = 'handheld'
    cfg['data']['scene'] = scene
    if rays:
        cfg['tracking']['pixels'] = rays
        cfg['mapping']['pixels'] = rays
    if iters_scale != 1.0:
        for sec, key in (('tracking', 'iters'), ('mapping', 'iters'), ('mapping', 'iters_first'), ('mapping', 'geo_iter_first')):
            cfg[sec][key] = max(1, int(round(cfg[sec][key] * iters_scale)))
    if iters_first is not None:
        cfg['mapping']['geo_iter_first'] = max(1, int(round(cfg['mapping']['geo_iter_first'] * iters_first / cfg['mapping']['iters_first'])))
        cfg['mapping']['iters_first'] = iters_first
    if seed is not None:
        cfg['setup_seed'] = seed
    if color_refine is not None:
        cfg['mapping']['color_refine'] = bool(color_refine)
    return cfg


def load_frames(cfg, device='cpu'):
    from loopy_slam_amd import slam
    rd = slam.SyntheticRoomDataset(cfg, device)
    return [rd[i] for i in range(len(rd))]


def summarise(est, gt, l1, extra):
    import oracle_slam as OS
    out = {'ate_rmse_cm': 100 * OS.ate_rmse(est, gt), 'rot_err_deg': OS.rotation_error_deg(est, gt),
           'depth_l1_cm': 100 * float(np.mean(l1)), 'depth_l1_cm_frames': [100 * x for x in l1],
           'max_translation_error_cm': 100 * float((est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1).max()),
           'path_length_cm': 100 * float((gt[1:, :3, 3] - gt[:-1, :3, 3]).norm(dim=1).sum()),
           'per_frame_motion_cm': 100 * float((gt[1:, :3, 3] - gt[:-1, :3, 3]).norm(dim=1).mean()),
           'prior_only': OS.prior_only_metrics(gt), 'est_c2w': est.tolist()}
    out.update(extra)
    return out


def run_oracle(cfg, threads):
    import oracle_slam as OS
    torch.set_num_threads(threads)
    frames = [tuple(x.cpu() if torch.is_tensor(x) else x for x in f) for f in load_frames(cfg)]
    o = OS.OracleSLAM(cfg, frames, log=lambda *a: print(*a, flush=True))
    t0 = time.time()
    est, gt = o.run()
    wall = time.time() - t0
    every = cfg['mapping']['every_frame']
    l1 = o.depth_l1(list(range(0, len(frames), every)))
    return summarise(est, gt, l1, {'pipeline': 'oracle (CPU restatement, torch autograd + torch.optim.Adam)', 'wall_s': round(wall, 1),
                                   'points': int(o.pos.shape[0]), 'keyframes': len(o.keyframe_list), 'map_log': o.map_log,
                                   'track_loss_first_best': o.track_log, 'threads': threads})


def product_depth_l1(ps, frames, est, ids, stride=4):
    """The same stride-4 pixel grid as the oracle's, through Renderer.render_batch_ray at the estimated poses."""
    from loopy_slam_amd.common import get_rays_from_uv
    from loopy_slam_amd import slam
    dev = ps.eng.device
    vals = []
    for k in ids:
        _, color, depth, _ = frames[k]
        jj, ii = torch.meshgrid(torch.arange(0, ps.H, stride, dtype=torch.float32, device=dev),
                                torch.arange(0, ps.W, stride, dtype=torch.float32, device=dev), indexing='ij')
        i, j = ii.reshape(-1), jj.reshape(-1)
        ro, rd = get_rays_from_uv(i, j, est[k].to(dev), ps.H, ps.W, ps.fx, ps.fy, ps.cx, ps.cy)
        gd = depth[j.long(), i.long()]
        rq = None
        if ps.cfg['use_dynamic_radius']:
            rq = torch.sqrt(slam.frame_radius_maps(ps.eng, ps.cfg, color)[2])[j.long(), i.long()]
        with torch.no_grad():
            d, _, _, _ = ps.renderer_map.render_batch_ray(ps.npc, ps.shared_decoders, rd.contiguous(), ro.contiguous(), dev, 'geometry',
                                                          gt_depth=gd, dynamic_r_query=rq)
        m = gd > 0
        vals.append(float((gd[m] - d[m]).abs().mean()))
    return vals


def run_product(cfg, emu=False):
    from loopy_slam_amd import slam
    eng = None
    if emu:
        from util import make_engine
        eng = make_engine('emu')
    torch.manual_seed(cfg.get('setup_seed', 1219))          # run.py's setup_seed: mlp_exposure's initial weights come from the global generator
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    frames = [ps.frame_reader[i] for i in range(len(ps.frame_reader))]

    class Pre:
        def __len__(self):
            return len(frames)

        def __getitem__(self, i):
            return frames[i]
    ps.frame_reader = Pre()
    sync = (lambda: None) if emu else torch.cuda.synchronize
    sync()
    t0 = time.time()
    est, gt = ps.run()
    sync()
    wall = time.time() - t0
    every = cfg['mapping']['every_frame']
    l1 = product_depth_l1(ps, frames, est, list(range(0, len(frames), every)))
    tl = ps.tracker.last_log
    return summarise(est, gt, l1, {'pipeline': 'product on the host emulator' if emu else 'product (libloopyhip on cuda:0)', 'wall_s': round(wall, 1),
                                   'points': ps.npc.pts_num(), 'keyframes': len(ps.mapper.keyframe_list),
                                   'track_loss_first_last': [float(tl[0, 0]), float(tl[-1, 0])] if tl is not None else None})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pipeline', choices=('oracle', 'hip', 'emu'), required=True)
    ap.add_argument('--config', default='configs/Synthetic/room.yaml')
    ap.add_argument('--frames', type=int, default=50)
    ap.add_argument('--rays', type=int, default=500)
    ap.add_argument('--iters-scale', type=float, default=1.0)
    ap.add_argument('--iters-first', type=int, default=None)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--color-refine', type=int, default=None)
    ap.add_argument('--scene', default='furnished', choices=('plain', 'furnished'))
    ap.add_argument('--threads', type=int, default=max(1, (os.cpu_count() or 2) // 2))
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    cfg = make_cfg(a.config, a.frames, a.rays, a.iters_scale, a.iters_first, a.seed, a.color_refine, a.scene)
    res = run_oracle(cfg, a.threads) if a.pipeline == 'oracle' else run_product(cfg, emu=a.pipeline == 'emu')
    res['config'] = {'file': a.config, 'frames': a.frames, 'rays_per_iteration': a.rays, 'iters_scale': a.iters_scale,
                     'tracking_iters': cfg['tracking']['iters'], 'mapping_iters': cfg['mapping']['iters'],
                     'iters_first': cfg['mapping']['iters_first'], 'seed': cfg.get('setup_seed', 1219),
                     'color_refine': bool(cfg['mapping'].get('color_refine', False)), 'motion': 'handheld', 'scene': a.scene}
    brief = {k: v for k, v in res.items() if k not in ('est_c2w', 'map_log', 'track_loss_first_best')}
    print(json.dumps(brief))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
