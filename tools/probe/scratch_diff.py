"""Which backward scratch buffer differs between two identical calls (debugging aid for the determinism test)."""
import sys, os
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
act1 = st.act.clone()
core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, 'color', save_act=True)
print('act repeat diff', float((st.act - act1).abs().max()), 'raw', )
d1, c1 = torch.randn(R, generator=g).cuda(), torch.randn(R, 3, generator=g).cuda()
gs = core.GradState(eng, pos.shape[0], R, blob.n, feats=True, weights=True)
core.render_backward(eng, st, gs, d1, c1); torch.cuda.synchronize()
s1 = gs.scratch.clone()
gs.zero_()
core.render_backward(eng, st, gs, d1, c1); torch.cuda.synchronize()
s2 = gs.scratch
P = R * 5
names = [('d_raw', 4 * P), ('dc_geo', 32 * P), ('dc_col', 32 * P), ('dp_embed', 4 * P), ('dp_embed_col', 4 * P), ('dp_rel', 4 * P), ('dp_total', 4 * P),
         ('dw_rel', 8 * P), ('w_eff', 8 * P), ('dlogit', 4 * P), ('part_bg', ((P + 31) // 32 + 3) // 4 * 288), ('part_br', ((P + 3) // 4 + 3) // 4 * 32),
         ('hbar', 128 * P), ('dfeat', 256 * P), ('w_sum', P), ('dh_col', 640 * P), ('rows', 1536 * P)]
o = 0
for n, sz in names:
    a, b = s1[o:o + sz], s2[o:o + sz]
    d = (a - b).abs()
    nz = int((d > 0).sum())
    msg = ''
    if nz and n in ('dc_col', 'dh_col'):
        w = 32 if n == 'dc_col' else 640
        rows = torch.nonzero(d.reshape(-1, w).amax(1) > 0).reshape(-1)
        cols = torch.nonzero(d.reshape(-1, w).amax(0) > 0).reshape(-1)
        msg = f' rows {rows[:8].tolist()} (#{rows.numel()}) tiles {sorted(set((rows // 32).tolist()))[:10]} cols {cols[:12].tolist()} (#{cols.numel()})'
    if nz and n == 'dh_col':
        D = d.reshape(-1, 640)
        t0 = int(rows[0]) // 32
        blk = D[t0 * 32:(t0 + 1) * 32] > 0
        print('   tile', t0, 'cols differing per 32-col block:', blk.any(0).reshape(20, 32).sum(1).tolist())
        print('   rows differing:', blk.any(1).int().tolist())
        cb = torch.nonzero(blk.any(0)).reshape(-1)
        print('   cols in block 16..19:', [int(c) for c in cb if c >= 512])
        print('   values run1', a.reshape(-1, 640)[t0 * 32, 512:520].tolist(), 'run2', b.reshape(-1, 640)[t0 * 32, 512:520].tolist())
    print(f'{n:14s} differing {nz:8d} max {float(d.max()) if sz else 0:.3e}{msg}')
    o += sz
