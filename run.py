#!/usr/bin/env python3
"""Command line of the reference's run.py (/root/reference/run.py:10-49) over the MI355X-native hot path:

    python run.py configs/Replica/room0.yaml [--input_folder DIR] [--output DIR] [--stop N]

builds loopy_slam_amd.slam.Point_SLAM (same constructor arguments: cfg, args, time_string) and runs it.  `--stop N` ends the run
after frame N with a checkpoint there (ckpt_freq = N, keyframe_every = 10, as the reference's deterministic-test mode does).
`--wandb / --no_wandb` are accepted and ignored (logging is out of scope).

Frames: the reference's dataset readers (src/utils/datasets.py: image decoding through cv2) are out of scope of this build and cv2 is
not installed; when `data.input_folder` is not a directory of frames this build can read, the synthetic 640x480 room
(loopy_slam_amd.synthetic) stands in at the config's intrinsics and the run says so.  A reader object with the reference's protocol
(`len`, `[i] -> (idx, color [H,W,3], depth [H,W], c2w [4,4])` on the device) can be handed to Point_SLAM(dataset=...).
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

from loopy_slam_amd import config
from loopy_slam_amd.slam import Point_SLAM, Logger


def setup_seed(seed):
    """src/common.py:32-38."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def main(argv=None):
    parser = argparse.ArgumentParser(description='Neural-point SLAM hot path on MI355X behind the reference command line.')
    parser.add_argument('config', type=str, help='scene configuration (YAML, inherit_from chain over configs/point_slam.yaml)')
    parser.add_argument('--input_folder', type=str, help='frames directory; overrides data.input_folder of the configuration')
    parser.add_argument('--output', type=str, help='results directory; overrides data.output of the configuration')
    parser.add_argument('--wandb', action='store_true')
    parser.add_argument('--no_wandb', action='store_true')

    def optional_int(string):
        return None if string == 'None' else int(string)
    parser.add_argument('--stop', type=optional_int, help='end the run after frame N (checkpoint there)')
    parser.add_argument('--frames', type=optional_int, default=None, help='(synthetic reader only) length of the sequence')
    args = parser.parse_args(argv)

    cfg = config.load_config(args.config, 'configs/point_slam.yaml')
    setup_seed(cfg['setup_seed'])
    if args.input_folder is not None:                       # Point_SLAM.py:54-63
        cfg['data']['input_folder'] = args.input_folder
    if args.output is not None:
        cfg['data']['output'] = args.output
    if args.stop:
        cfg['mapping']['ckpt_freq'] = args.stop
        cfg['mapping']['keyframe_every'] = 10
    if args.frames:
        cfg['data']['n_frames'] = args.frames
    elif args.stop and 'n_frames' not in cfg['data']:
        cfg['data']['n_frames'] = args.stop + 1

    from datetime import datetime
    time_string = datetime.now().strftime('%Y%m%d_%H%M%S') if args.stop is None else None

    # Dataset readers (src/utils/datasets.py) are out of scope of this build: whatever data.input_folder says - an existing directory
    # included - the frames come from the synthetic room, and every printed error is against ITS ground truth.  Said on every run.
    folder = cfg['data'].get('input_folder')
    print(f'run.py: NO FRAME READER in this build - data.input_folder = {folder!r} '
          f'({"exists, NOT read" if (folder and os.path.isdir(folder)) else "not found"}): the synthetic room stands in '
          f'({cfg["data"].get("n_frames", 50)} frames at the config\'s intrinsics); pass a dataset object to Point_SLAM(dataset=...) for real frames',
          flush=True)
    slam = Point_SLAM(cfg, args, time_string=time_string)
    slam.mapper.logger = Logger(cfg, args, slam.mapper)     # checkpoints in the reference's layout under <output>/ckpts (Point_SLAM.py:126-132)
    est, gt = slam.run()
    n = est.shape[0]
    err = (est[:, :3, 3] - gt[:, :3, 3]).norm(dim=1)
    print(f'run.py: {n} frames, {slam.npc.pts_num()} neural points, {len(slam.mapper.keyframe_list)} keyframes, '
          f'mean |t - t_gt| = {100 * float(err.mean()):.2f} cm; checkpoints in {slam.mapper.logger.ckptsdir}', flush=True)
    return slam


if __name__ == '__main__':
    main()
