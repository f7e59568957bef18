"""Shared helpers for tests (golden loading, weight dicts, tolerances)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CFG_NAMES = ('replica', 'tum', 'scannet')


def load(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def tens(d, *keys):
    return [torch.from_numpy(np.asarray(d[k])) for k in keys]


def weights(name):
    return {k: torch.from_numpy(v) for k, v in load('weights_' + name).items()}


# resolved hot-path knobs per dataset config
CFG = {
    'replica': dict(near_surface=0.98, far_surface=1.02, rel_pos=True, exposure=False, dynamic=False),
    'tum': dict(near_surface=0.98, far_surface=1.02, rel_pos=False, exposure=False, dynamic=True),
    'scannet': dict(near_surface=0.96, far_surface=1.04, rel_pos=False, exposure=True, dynamic=True),
}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ---------------------------------------------------------------------------- kernel back-ends
_EMU = {}


def make_engine(backend):
    """backend 'hip': the product library on cuda:0 (GPU tests).
    backend 'emu': TEST-ONLY host build of the same .hip sources against tests/hipemu (CPU tests of
    the kernels' logic); the product package itself never loads it."""
    import sys
    import torch
    from loopy_slam_amd import _ffi, core
    if backend == 'hip':
        return core.Engine()
    if 'lib' not in _EMU:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))
        import build_emu
        # LOOPY_EMU_LIB: another build of the same emulator sources (e.g. with -fsanitize=address, tools/emu_asan.sh)
        _EMU['lib'] = _ffi.LoopyLib(os.environ.get('LOOPY_EMU_LIB') or build_emu.build())
    return core.Engine(lib=_EMU['lib'], device='cpu')


def backends():
    import pytest
    return [pytest.param('emu', id='emu'), pytest.param('hip', marks=pytest.mark.gpu, id='hip')]
