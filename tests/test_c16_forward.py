"""The decoder forward in its 16 x 16 x 32 matrix-instruction form (k_decode_fwd16, lk_debug_set_c16 / LK_C16=1; off by default - measured
slower on MI355X): the reference-run goldens of the three configs, mapper and tracker mode, zero-depth rays, through
BOTH forms; the saved activation rows of the two forms are what the (unchanged) backward reads, so a training iteration agrees too."""
import numpy as np
import pytest
import torch

from util import CFG_NAMES, CFG, load, make_engine, backends
import test_forward_parity as TF
import test_backward_parity as TB


@pytest.fixture
def c16(request):
    eng = make_engine(request.param)
    eng.lib.check(eng.lib.dll.lk_debug_set_c16(1), 'lk_debug_set_c16')
    yield eng
    eng.lib.check(eng.lib.dll.lk_debug_set_c16(0), 'lk_debug_set_c16')


@pytest.mark.parametrize('c16', backends(), indirect=True)
@pytest.mark.parametrize('stage', ('geometry', 'color'))
@pytest.mark.parametrize('name', CFG_NAMES)
def test_forward_goldens_in_the_16x16_form(c16, name, stage):
    g = load(f'g6_render_{name}_map_{stage}')
    st = TF.run_forward(c16, name, g, stage, color_logits=CFG[name]['exposure'])
    TF.check_outputs(st, g)
    # against the default form: the same products in another association order
    c16.lib.dll.lk_debug_set_c16(0)
    st0 = TF.run_forward(c16, name, g, stage, color_logits=CFG[name]['exposure'])
    c16.lib.dll.lk_debug_set_c16(1)
    assert float((st.raw - st0.raw).abs().max()) < 2e-5
    np.testing.assert_allclose(st.depth.cpu().numpy(), st0.depth.cpu().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('c16', backends(), indirect=True)
@pytest.mark.parametrize('name', CFG_NAMES)
def test_backward_goldens_on_rows_saved_by_the_16x16_form(c16, name):
    """lk_render_bwd (32 x 32 form) on the activation rows the 16 x 16 forward saved: every gradient of the reference-autograd goldens,
    mapper (colour stage, unit-scale loss gradients) and tracker mode."""
    backend = 'emu' if c16.device.type == 'cpu' else 'hip'
    TB.test_backward_mapper_golden(backend, name, 'color', True, False)
    TB.test_backward_tracker_golden(backend, name)
