import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _poison_uninitialised_buffers():
    """Every buffer the product allocates WITHOUT initialising it (core.Engine.empty: scratch, work buffers, outputs) starts as NaN in the
    tests (bytes 0xA5 / ints 0x3A3A3A3A on the host emulator; floats only on the GPU, where a wild index would fault the box; buffers above 256 MB
    are left alone).  A kernel that reads what nobody wrote then fails a test deterministically instead of depending on what the allocator
    hands back: round 3 found two such reads this way (dead samples of a partitioned batch) after ONE run of the CPU suite had
    failed and three had passed - freed -1 index tables are NaN bit patterns."""
    import torch
    from loopy_slam_amd import core
    if getattr(core.Engine, '_poisoned', False):
        return
    orig = core.Engine.empty

    def empty(self, *shape, dtype=torch.float32):
        t = orig(self, *shape, dtype=dtype)
        if t.numel() * t.element_size() <= (256 << 20):
            if t.dtype.is_floating_point:
                t.fill_(float('nan'))
            elif t.device.type == 'cpu':
                t.fill_(0xA5 if t.dtype == torch.uint8 else (0x3A3A3A3A if t.dtype == torch.int32 else 7))
        return t
    core.Engine.empty = empty
    core.Engine._poisoned = True


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('LOOPY_NO_POISON') != '1':
        _poison_uninitialised_buffers()


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically when no GPU is visible (the CPU container).
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
