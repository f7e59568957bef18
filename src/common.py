"""`from src.common import setup_seed, get_camera_from_tensor, ...` (run.py:9, src/common.py)."""
import random

import numpy as np
import torch

from loopy_slam_amd.common import (get_camera_from_tensor, get_tensor_from_camera, get_rays, get_rays_from_uv, get_samples,  # noqa: F401
                                   quad2rotation)


def setup_seed(seed):
    """Seeds of torch, numpy and random (src/common.py:32-38)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
