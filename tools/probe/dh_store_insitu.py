"""In-situ side of the store-data investigation: the colour decoder backward built with
-DLK_DH_STORE_MODE=m (lk_bwd.hip) - 0: d h stored straight from the accumulators, 2: + 64 idle cycles after the stores,
3: + 64 idle cycles before the stores, 4: + s_waitcnt vmcnt(0) after the stores, 1 (product): register copy held to the
end of the layer.  For every build: the same backward four times at R rays, rows of d h / d c that differ between
repeats, and which sample columns of the 32-sample tiles they sit in.

    bash tools/ab_build.sh dh0 -DLK_DH_STORE_MODE=0   (... dh2, dh3, dh4)      # in the build container
    python tools/probe/dh_store_insitu.py [R]                                      # on the GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from loopy_slam_amd import _ffi, core, synthetic as syn

I = syn.TUM_INTR


def cdiv(a, b):
    return (a + b - 1) // b


def run(lib_path, R, unit):
    eng = core.Engine(lib=_ffi.LoopyLib(lib_path))
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
    xf = 0
    if unit:
        xf = _ffi.FLAG_UNIT_LOSS_GRADS
        d1, c1 = torch.sign(d1), 0.1 * torch.sign(c1)
    core.render_forward(eng, cfg, st, ro, rd, gd, knn, pos, geo, col, blob, 'color', save_act=True, extra_flags=xf)
    P = R * cfg.S
    o_dc = (4 + 32) * P
    o_dh = (4 + 32 + 32 + 4 + 4 + 4 + 4 + 8 + 8 + 4) * P + cdiv(cdiv(P, 32), 4) * 288 + cdiv(cdiv(P, 4), 4) * 32 + 128 * P + 256 * P + P
    ref, bad_dc, bad_dh, cols = None, 0, 0, torch.zeros(32, dtype=torch.long)
    for rep in range(4):
        gs = core.GradState(eng, pos.shape[0], R, blob.n, feats=True, weights=True)
        core.render_backward(eng, st, gs, d1, c1)
        torch.cuda.synchronize()
        cur = (gs.scratch[o_dc:o_dc + 32 * P].clone(), gs.scratch[o_dh:o_dh + 640 * P].clone())
        if ref is None:
            ref = cur
        else:
            bad_dc += int((cur[0] != ref[0]).reshape(-1, 32).any(1).sum())
            rows = (cur[1] != ref[1]).reshape(-1, 640).any(1)
            bad_dh += int(rows.sum())
            cols += torch.bincount(torch.nonzero(rows).reshape(-1).cpu() % 32, minlength=32)
    return bad_dc, bad_dh, cols.tolist()


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    libs = [('product (mode 1: held copy)', _ffi.LIB_PATH)]
    for m, what in ((0, 'no copy'), (2, 'no copy + 64 idle cycles AFTER the stores'), (3, 'no copy + 64 idle cycles BEFORE the stores'),
                    (4, 'no copy + s_waitcnt vmcnt(0) after the stores')):
        p = os.path.join(ROOT, 'ab', f'lib_dh{m}.so')
        if os.path.exists(p):
            libs.append((f'mode {m}: {what}', p))
    print(f'# d h / d c rows differing between 4 repeats of the same backward, R = {R} rays, N = 100 000 points')
    for name, path in libs:
        for unit in (False, True):
            dc, dh, cols = run(path, R, unit)
            lo, hi = sum(cols[:16]), sum(cols[16:])
            print(f'{name:58s} {"fp16 pieces (unit grads)" if unit else "bf16 pieces":26s} d c rows {dc:6d}  d h rows {dh:6d}'
                  f'  (sample columns 0-15: {lo}, 16-31: {hi})', flush=True)


if __name__ == '__main__':
    main()
