"""Import the (Python) reference from /root/reference inside THIS container only.

Used exclusively by tools/gen_golden.py to produce the golden vectors committed
under tests/golden/.  Nothing here (and nothing under /root/reference) travels to
the GPU box; tests, smoke() and bench.py never import this module.

The reference needs third-party modules that are not installed here (faiss-gpu,
open3d, cv2, skimage, pydbow3, wandb, ...).  None of them is touched by the
functions we capture, so each is replaced by an empty permissive stub module
(SURVEY.md Appendix D).
"""
import sys
import types
import contextlib

import torch  # noqa: F401  (import first so torch's own inspect calls never see stubs)

REFERENCE_ROOT = '/root/reference'

_STUBS = [
    'cv2', 'open3d', 'open3d.core', 'skimage', 'skimage.color', 'skimage.filters',
    'turtle', 'faiss', 'faiss.contrib', 'faiss.contrib.torch_utils', 'pydbow3',
    'wandb', 'colorama', 'torchmetrics', 'torchmetrics.image',
    'torchmetrics.image.lpip', 'pytorch_msssim',
]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Dummy()


def _make_stub(name):
    m = types.ModuleType(name)

    def _getattr(attr):
        if attr.startswith('__'):
            raise AttributeError(attr)
        return _Dummy()
    m.__getattr__ = _getattr
    m.__path__ = []
    return m


def import_reference():
    """Return a namespace with the reference modules used for fixture generation."""
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = _make_stub(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import src.common as common
    import src.conv_onet.models.decoder as decoder
    import src.utils.Renderer as renderer
    import src.config as config
    return types.SimpleNamespace(common=common, decoder=decoder,
                                 renderer=renderer, config=config)


@contextlib.contextmanager
def cpu_get_device_patch():
    """quad2rotation does `.to(quad.get_device())` which is -1 on CPU
    (reference src/common.py:314); scope-patch get_device to return 'cpu'."""
    orig = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: 'cpu'
    try:
        yield
    finally:
        torch.Tensor.get_device = orig


def load_cfg(ref, rel_path):
    import os
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    try:
        return ref.config.load_config(rel_path, 'configs/point_slam.yaml')
    finally:
        os.chdir(cwd)
