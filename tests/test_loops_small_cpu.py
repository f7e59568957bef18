"""tests/test_loops_at_size.py (`-m gpu`: the native loops against the oracle loop at the benchmark's sizes) on the host emulator at a
reduced size: the tests' own plumbing - oracle loops with the KD-tree search, the gradient-pixel pool and dynamic radii handed over as
inputs, the trajectory bounds - runs in the CPU suite too."""
import pytest
import torch

import atsize as A
import test_loops_at_size as L
from util import make_engine


@pytest.fixture
def small(monkeypatch):
    scene0 = A.scene
    monkeypatch.setattr(A, 'scene', lambda N, seed=1219: scene0(min(N, 24_000), seed))
    torch.set_num_threads(4)
    yield
    L._REPORT.clear()


def test_track_call_small(small):
    L.run_track_case(make_engine('emu'), 'cpu-track-replica', 24_000, 128, 8, True, True, 0.002)
    assert max(L._REPORT['cpu-track-replica-stiff']['loss_rel']) <= 5e-5


def test_track_call_small_gradient_pool(small):
    L.run_track_case(make_engine('emu'), 'cpu-track-tum', 24_000, 160, 4, False, False, 0.002, grad_pool=True, dynamic=True)


def test_map_call_small(small):
    L.run_map_case(make_engine('emu'), 'cpu-map-replica', 24_000, 160, 7, 3, True, window=4)
