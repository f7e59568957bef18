"""Diagnosis of the round-1 'store-data' failure on the HISTORICAL source (commit 2401a79 exported to ab/hist_*, the d h
register copy removed in ab/hist_nocopy): run inside one of those trees,

    cd ab/hist_nocopy && python ../../tools/probe/hist_diag.py save /tmp/ref.pt      (in ab/hist_copy: the reference bits)
    cd ab/hist_nocopy && python ../../tools/probe/hist_diag.py cmp /tmp/ref.pt

Reports which (layer, wave block) tiles of d h and d c differ between repeats and from the reference build, in which sample
columns of the 32-sample tile, and what the wrong values are (another tile's values? zeros? scaled?)."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from loopy_slam_amd import core, synthetic as syn

I = syn.TUM_INTR


def cdiv(a, b):
    return (a + b - 1) // b


def run(R=10000, reps=3):
    eng = core.Engine()
    pos, geo, col = syn.build_cloud(100_000, device='cpu')
    pos, geo, col = eng.f32(pos), eng.f32(geo), eng.f32(col)
    knn = core.KnnIndex(eng, capacity=pos.shape[0])
    knn.build(pos)
    blob = core.DecoderBlob(eng).pack(syn.default_weights())
    cfg = core.RenderCfg()
    depth, _, c2w = syn.render_frame(7, device='cuda', holes=0.02)
    g = torch.Generator().manual_seed(R)
    i = torch.randint(0, I['W'], (R,), generator=g).float().cuda()
    j = torch.randint(0, I['H'], (R,), generator=g).float().cuda()
    ro, rd = syn.pixel_rays(c2w, i, j)
    gd = depth[j.long(), i.long()].contiguous()
    st = core.RenderState(eng, R, cfg.S, need_act=True)
    d1, c1 = torch.randn(R, generator=g).cuda(), torch.randn(R, 3, generator=g).cuda()
    P = R * cfg.S
    o_dc = (4 + 32) * P
    o_dh = (4 + 32 + 32 + 4 + 4 + 4 + 4 + 8 + 8 + 4) * P + cdiv(cdiv(P, 32), 4) * 288 + cdiv(cdiv(P, 4), 4) * 32 + 128 * P + 256 * P + P
    outs = []
    for rep in range(reps):
        core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, 'color', save_act=True)
        gs = core.GradState(eng, pos.shape[0], R, blob.n, feats=True, weights=True)
        core.render_backward(eng, st, gs, d1, c1)
        torch.cuda.synchronize()
        outs.append((gs.scratch[o_dc:o_dc + 32 * P].reshape(P, 32).cpu().clone(), gs.scratch[o_dh:o_dh + 640 * P].reshape(P, 640).cpu().clone()))
    return outs


def describe(tag, got, ref):
    dc_g, dh_g = got
    dc_r, dh_r = ref
    P = dh_g.shape[0]
    bad_dc = (dc_g != dc_r).any(1)
    print(f'{tag}: d c rows differing {int(bad_dc.sum())} of {P}')
    blk = (dh_g != dh_r).reshape(P, 5, 4, 32).any(3)              # [P, layer, wave]
    print('  d h differing (layer i, wave w) -> rows:', {(i, w): int(blk[:, i, w].sum()) for i in range(5) for w in range(4) if int(blk[:, i, w].sum())})
    rows = torch.nonzero(blk.any(2).any(1)).reshape(-1)
    if rows.numel():
        cols = torch.bincount(rows % 32, minlength=32)
        print('  sample columns 0-15:', int(cols[:16].sum()), ' 16-31:', int(cols[16:].sum()), ' tiles hit:', int(torch.unique(rows // 32).numel()), 'of', cdiv(P, 32))
        # what are the wrong values?  look at the first few (row, layer, wave) blocks
        shown = 0
        for r in rows.tolist():
            for i in range(5):
                for w in range(4):
                    if blk[r, i, w] and shown < 6:
                        g, f = dh_g[r, i * 128 + w * 32:i * 128 + w * 32 + 32], dh_r[r, i * 128 + w * 32:i * 128 + w * 32 + 32]
                        ratio = (g / f)[f.abs() > 1e-12]
                        same_as = [(ii, ww) for ii in range(5) for ww in range(4) if (ii, ww) != (i, w) and torch.equal(g, dh_r[r, ii * 128 + ww * 32:ii * 128 + ww * 32 + 32])]
                        nd = int((g != f).sum())
                        print(f'    row {r} (tile {r // 32}, column {r % 32}) layer {i} wave {w}: {nd}/32 elements differ; got[:4] {g[:4].tolist()} ref[:4] {f[:4].tolist()} '
                              f'ratio median {float(ratio.median()) if ratio.numel() else float("nan"):.4g}; equals another block of the row: {same_as}; which elements: {torch.nonzero(g != f).reshape(-1).tolist()[:16]}')
                        shown += 1


if __name__ == '__main__':
    mode, path = sys.argv[1], sys.argv[2]
    outs = run()
    if mode == 'save':
        torch.save(outs[0], path)
        for k in range(1, len(outs)):
            describe(f'reference build, repeat {k} vs repeat 0', outs[k], outs[0])
    else:
        ref = torch.load(path)
        for k in range(len(outs)):
            describe(f'repeat {k} vs reference build', outs[k], ref)
        describe('repeat 1 vs repeat 0 (same build)', outs[1], outs[0])
