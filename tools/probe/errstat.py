"""Forward / gradient error of the kernels against the reference goldens (max abs, relative to max |g| for gradients)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import test_backward_parity as T
from util import make_engine
from loopy_slam_amd import core
from oracle import hotpath as H
eng = make_engine(sys.argv[1] if len(sys.argv) > 1 else 'hip')
for name in ('replica', 'tum'):
    g = T.load(f'g6_render_{name}_map_color')
    cfg, dec, st, N, R = T.setup(eng, name, g, 'color', color_logits=T.CFG[name]['exposure'])
    depth = st.depth.cpu().clone().requires_grad_(True); color = st.color.cpu().clone().requires_grad_(True)
    gd, gc = T.tens(g, 'gt_depth', 'gt_color')
    loss, _, _, _ = H.mapper_loss(depth, color, st.valid_ray.cpu().bool(), gd, gc, 'color', float(g['w_color']))
    loss.backward()
    gs = core.GradState(eng, N, R, dec.n, feats=True, weights=True)
    core.render_backward(eng, st, gs, eng.f32(depth.grad), eng.f32(color.grad))
    print(name, 'fwd depth', float(np.abs(st.depth.cpu().numpy() - g['depth']).max()), 'color', float(np.abs(st.color.cpu().numpy() - g['color']).max()),
          'var', float(np.abs(st.var.cpu().numpy() - g['var']).max()) if 'var' in g else '')
    print(name, 'g_geo', T.relerr(gs.g_geo.cpu(), g['grad_geo']), 'g_col', T.relerr(gs.g_col.cpu(), g['grad_col']))
    gW = dec.unpack(gs.g_weights); worst = 0
    for k, gv in g.items():
        if k.startswith('gradW.') and k[6:] in gW and not (k[6:].startswith('geo_decoder.') and k[6:] != 'geo_decoder.embedder._B'):
            worst = max(worst, T.relerr(gW[k[6:]].reshape(gv.shape), gv))
    print(name, 'worst gradW', worst)
