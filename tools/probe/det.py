"""Bitwise repeatability of d c / d h of the colour decoder backward (debugging aid)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from loopy_slam_amd import core, synthetic as syn
import test_fullsize_gpu as T
eng = T.make_engine('hip')
pos, geo, col, knn = T._scene(eng, 100_000)
blob = core.DecoderBlob(eng).pack(syn.default_weights())
cfg = core.RenderCfg()
depth, _, c2w = syn.render_frame(7, device='cuda', holes=0.02)
R = 10000
g = torch.Generator().manual_seed(1)
i = torch.randint(0, T.I['W'], (R,), generator=g).float().cuda(); j = torch.randint(0, T.I['H'], (R,), generator=g).float().cuda()
ro, rd = syn.pixel_rays(c2w, i, j); gd = depth[j.long(), i.long()].contiguous()
st = core.RenderState(eng, R, cfg.S, need_act=True)
core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, 'color', save_act=True)
d1, c1 = torch.randn(R, generator=g).cuda(), torch.randn(R, 3, generator=g).cuda()
P = R * 5
o_dc = (4 + 32) * P
o_dh = (4 + 32 + 32 + 4 + 4 + 4 + 4 + 8 + 8 + 4) * P + ((P + 31) // 32 + 3) // 4 * 288 + ((P + 3) // 4 + 3) // 4 * 32 + 128 * P + 256 * P + P
ref = None; bad_dc = bad_dh = 0
for rep in range(4):
    gs = core.GradState(eng, pos.shape[0], R, blob.n, feats=True, weights=True)
    core.render_backward(eng, st, gs, d1, c1); torch.cuda.synchronize()
    cur = (gs.scratch[o_dc:o_dc + 32 * P].clone(), gs.scratch[o_dh:o_dh + 640 * P].clone())
    if ref is None: ref = cur
    else:
        bad_dc += int((cur[0] != ref[0]).reshape(-1, 32).any(1).sum()); bad_dh += int((cur[1] != ref[1]).reshape(-1, 640).any(1).sum())
print('rows differing over 3 repeats: dc', bad_dc, 'dh', bad_dh)
