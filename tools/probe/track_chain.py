#!/usr/bin/env python3
"""The dependent chain of ONE tracking iteration, phase by phase, on one time axis (round-5 review item 3).

Needs the probe build (tools/ab_build.sh chain -DLK_PROBE_CHAIN, copied over loopy_slam_amd/libloopyhip.so): the four launches of an
iteration - k_sample_interp_pose, k_relpos_decode_fwd, k_decode_bwd, k_relpos_interp_bwd - leave 100-MHz wall-clock stamps
(s_memrealtime: one clock for all compute units and launches, 10 ns a tick) at their phase boundaries, per workgroup and wave.  The
buffers hold the LAST iteration of the call.

    python tools/probe/track_chain.py [iterations] > profiles/r6_track_chain.md

Per launch: when its workgroups start and end relative to the iteration's first stamp, the median workgroup's progress through the
phases, and the workgroup that ends last (the one the next launch waits for).  W = the stamp waits for the wave's outstanding memory
operations first (the data HAS arrived), so the phases are the dependent round trips of the chain.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from loopy_slam_amd import core, workload

WGS, WAVES, SLOTS = 512, 8, 16
TICK_US = 0.01

K1 = ('k_sample_interp_pose<16>', 'sample', [
    (0, 'entry'), (1, 'pose stepped (partials reduced, Adam, LDS)'), (2, 'W query point known (reading, pixel)'), (3, 'W row table arrived (cell_start)'),
    (4, 'W first 16 candidates of every row ranked'), (5, 'W remaining candidates ranked'), (6, 'lists merged over the 16 lanes'),
    (7, 'search returned'), (8, 'weights normalised'), (9, 'W feature rows arrived and summed'), (10, 'W stores retired')])
K2C = ('k_relpos_decode_fwd (colour waves 0-3)', 'fwd', [
    (0, 'entry'), (1, 'W rel-pos inputs arrived (lists, positions, rows)'), (2, 'rel-pos hidden layer'), (3, 'rel-pos output, c_col rows stored'),
    (4, 'workgroup barrier'), (14, 'W decoder set-up (sample, embedding share, c_col)'), (5, 'barrier 0'), (6, 'layer 0 + barrier'), (7, 'layer 1 + barrier'),
    (8, 'layer 2 + barrier'), (9, 'layer 3 + barrier'), (10, 'layer 4'), (11, 'output partials + barrier'), (12, 'wave 0: geometry wave done'),
    (13, 'W wave 0: composite, outputs stored')])
K2G = ('k_relpos_decode_fwd (geometry wave 4)', 'fwd', [
    (0, 'entry'), (1, 'W rel-pos inputs arrived'), (2, 'rel-pos hidden layer'), (3, 'rel-pos output stored'), (4, 'workgroup barrier'),
    (5, 'W embedding (96 sin) + c_geo'), (6, 'layer 0'), (7, 'layer 1'), (8, 'layer 2'), (9, 'layer 3'), (10, 'layer 4 + output')])
# the forward as ONE launch (k_track_fwd, round 6): pose step, search, rel-pos MLP, decoders
KFC = ('k_track_fwd (colour waves 0-3)', 'fwd', [
    (0, 'entry'), (3, 'pose stepped (partials reduced, Adam, LDS)'), (4, 'search + interpolation of the wave\'s four samples (lists in registers, stores issued)'),
    (1, 'W rel-pos operands arrived (positions, rows)'), (2, 'rel-pos hidden layer'), (15, 'rel-pos output + workgroup barrier'),
    (14, 'W embedding share, c_col from LDS'), (5, 'barrier 0'), (6, 'layer 0 + barrier'), (7, 'layer 1 + barrier'),
    (8, 'layer 2 + barrier'), (9, 'layer 3 + barrier'), (10, 'layer 4'), (11, 'output partials + barrier'), (12, 'wave 0: geometry wave done'),
    (13, 'W wave 0: composite, outputs stored')])
KFG = ('k_track_fwd (geometry wave 4)', 'fwd', [
    (0, 'entry'), (3, 'pose stepped'), (4, 'search + interpolation'), (1, 'W rel-pos operands arrived'), (2, 'rel-pos hidden layer'),
    (15, 'rel-pos output + workgroup barrier'), (5, 'W embedding (96 sin), c_geo from LDS'), (6, 'layer 0'), (7, 'layer 1'), (8, 'layer 2'), (9, 'layer 3'),
    (10, 'layer 4 + output')])
K3C = ('k_decode_bwd<true, true, false> (colour tiles)', 'bwd', [
    (0, 'entry'), (1, 'W threshold, loss term, composite backward'), (2, 'W d h_4, layer 4 operands'), (3, 'layer 4 parked + barrier'),
    (4, 'layer 3'), (5, 'layer 2'), (6, 'layer 1'), (7, 'layer 0'), (8, 'embedding gradient, d c partials + barrier'), (9, 'W stores retired')])
K3G = ('k_decode_bwd<true, true, false> (geometry waves)', 'bwd', [
    (0, 'entry'), (1, 'W threshold, loss term, composite backward'), (3, 'layer 4'), (4, 'layer 3'), (5, 'layer 2'), (6, 'layer 1'), (7, 'layer 0'),
    (8, 'd c_geo stored'), (9, 'W embedding gradient, stores retired')])
K4 = ('k_relpos_interp_bwd<false>', 'bwd2', [
    (0, 'entry'), (1, 'W rel-pos inputs arrived'), (2, 'hidden layer recomputed'), (3, 'd hid'), (4, 'd x'), (5, 'rel-pos rows stored'),
    (6, 'workgroup barrier'), (7, 'W interpolation: lists, d c, geometry rows -> d weight'), (8, 'W positions -> d p'), (9, 'W pose partials stored')])


def read(eng, name):
    fn = getattr(eng.lib.dll, 'lk_debug_chain_' + name)
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    fn.restype = C.c_int
    buf = (C.c_ulonglong * (WGS * WAVES * SLOTS))()
    assert fn(buf) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(WGS, WAVES, SLOTS).astype(np.int64)


def table(title, t, t0, wg_sel, waves, slots, out):
    """t: [wg][wave][slot]; rows = slots; columns = progress of the median / slowest workgroup (max over the selected waves)"""
    sel = t[wg_sel][:, waves, :]                    # [n_wg][n_wave][slot]
    sl = [s for s, _ in slots]
    ok = sel[:, :, sl[0]] > 0
    prog = np.where(ok[:, :, None], sel[:, :, sl], 0).max(axis=1).astype(np.float64)         # per workgroup: its last wave at each slot
    prog = prog[prog[:, 0] > 0]
    rel = (prog - t0) * TICK_US
    last = int(np.argmax(rel[:, -1]))
    out.append(f'### {title}: {rel.shape[0]} workgroups\n')
    out.append('| phase boundary | median workgroup, us from the iteration\'s start | phase, us (median) | workgroup that ends last | its phase, us |')
    out.append('|---|---|---|---|---|')
    prev_m, prev_l = None, None
    for i, (s, label) in enumerate(slots):
        m, l = float(np.median(rel[:, i])), float(rel[last, i])
        dm = '' if prev_m is None else f'{np.median(rel[:, i] - rel[:, i - 1]):.2f}'
        dl = '' if prev_l is None else f'{l - prev_l:.2f}'
        out.append(f'| {label} | {m:.2f} | {dm} | {l:.2f} | {dl} |')
        prev_m, prev_l = m, l
    e0, e1 = rel[:, 0], rel[:, -1]
    out.append(f'\nentries {e0.min():.2f} .. {e0.max():.2f} us (median {np.median(e0):.2f}); ends {e1.min():.2f} .. {e1.max():.2f} us (median {np.median(e1):.2f}); '
               f'median workgroup lives {np.median(e1 - e0):.2f} us\n')
    return float(e0.min()), float(e1.max())


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    eng = core.Engine()
    b = workload.Budget()
    wl = workload.FrameWorkload(eng, b)
    H, W = wl.H, wl.W
    e = min(b.ignore_edge, H // 4)
    win = (e, H - e, e, W - e)
    rnd = wl._draws(iters, b.track_rays, (win[1] - win[0]) * (win[3] - win[2]))
    for _ in range(3):
        wl.tracker.track(wl.cam0, wl.depth_stack[0], wl.color_stack[0], iters, win, wl.intr, rnd)
    torch.cuda.synchronize()
    t = {n: read(eng, n) for n in ('sample', 'fwd', 'bwd', 'bwd2')}
    R, S = b.track_rays, wl.cfg.S
    P = R * S
    n_k1 = (P + 15) // 16
    ts = (32 // S) * S
    n_k2 = (P + ts - 1) // ts
    tiles = (P + 31) // 32
    n_geo = (tiles + 3) // 4
    merged = not (t['sample'][:n_k1, :4, 0] > 0).any()          # k_track_fwd: the search is part of the forward launch
    t0 = t['fwd'][:n_k2, :4, 0] if merged else t['sample'][:n_k1, :4, 0]
    t0 = int(t0[t0 > 0].min())
    out = [f'# One tracking iteration as a dependent chain (round 6): wall-clock stamps of the LAST of {iters} iterations, {R} rays x {S} samples, '
           f'N = {b.n_points} points', '',
           '`python tools/probe/track_chain.py` on the probe build (`tools/ab_build.sh chain -DLK_PROBE_CHAIN`): every wave stamps s_memrealtime (100 MHz, one clock',
           'for the chip) at its phase boundaries; W = after `s_waitcnt 0`, i.e. the data the phase asked for HAS arrived.  The stamps cost the',
           'launches 1-2 us each (the waits serialise loads the shipped build overlaps); read the table for the SHAPE of the chain, the shipped',
           'durations are in `r6_iteration_timeline.md`.  Times in us from the first stamp of the iteration\'s first launch.', '']
    spans = []
    if merged:
        a = table(KFC[0], t['fwd'], t0, slice(0, n_k2), [0, 1, 2, 3], KFC[2], out)
        g = table(KFG[0], t['fwd'], t0, slice(0, n_k2), [4], KFG[2], out)
        spans.append(('k_track_fwd', (min(a[0], g[0]), max(a[1], g[1]))))
    else:
        spans.append(('k_sample_interp_pose', table(K1[0], t['sample'], t0, slice(0, n_k1), [0, 1, 2, 3], K1[2], out)))
        a = table(K2C[0], t['fwd'], t0, slice(0, n_k2), [0, 1, 2, 3], K2C[2], out)
        g = table(K2G[0], t['fwd'], t0, slice(0, n_k2), [4], K2G[2], out)
        spans.append(('k_relpos_decode_fwd', (min(a[0], g[0]), max(a[1], g[1]))))
    a = table(K3C[0], t['bwd'], t0, slice(n_geo, n_geo + tiles), [0, 1, 2, 3], K3C[2], out)
    # a geometry workgroup is four independent one-wave tiles: a "workgroup" row here is the slowest of its waves
    g = table(K3G[0], t['bwd'], t0, slice(0, n_geo), [0, 1, 2, 3], K3G[2], out)
    spans.append(('k_decode_bwd', (min(a[0], g[0]), max(a[1], g[1]))))
    spans.append(('k_relpos_interp_bwd', table(K4[0], t['bwd2'], t0, slice(0, tiles), [0, 1, 2, 3], K4[2], out)))
    out.append('## The four launches on the axis\n')
    out.append('| launch | first entry | last end | span, us | gap to the launch before (last end -> first entry), us |')
    out.append('|---|---|---|---|---|')
    prev = None
    for n, (s0, s1) in spans:
        out.append(f'| {n} | {s0:.2f} | {s1:.2f} | {s1 - s0:.2f} | {"" if prev is None else f"{s0 - prev:.2f}"} |')
        prev = s1
    print('\n'.join(out))


if __name__ == '__main__':
    main()
