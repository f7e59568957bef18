"""Per-kernel timing (lk_profile_*: HIP events on the launch stream) and the roofline bookkeeping
bench.py reports.  Peaks from /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32)
157.3 TFLOP/s dense, HBM3E 8 TB/s.

Algorithmic work per unit (DESIGN.md §Kernels states the same figures):
  multiply-adds per SAMPLE POINT
    decode fwd   geometry 15 479 - 279 (embedding on the VALU) = 15 200 ; colour 96 640
    decode bwd   (backward-data) geometry 15 392 ; colour 86 400 (+ 10 240 embedding columns in tracker mode)
    rel-pos fwd  8 neighbours x (128x52 + 32x128) = 86 016 ;  bwd 8 x (128x52 fwd + 32x128 + 128x52 bwd) = 139 264
                 (+ 8 x 32x128 for the recomputed output in tracker mode)
    wgrad        colour matrices 92 640 (k_wgrad: W_0..W_4, Wo and the 32 auxiliary columns M_i = d y_i^T c of four trunk jobs and the
                 output job, from which the reduction launch forms the fc_c gradients); the rel-pos matrices ride in
                 k_relpos_bwd_fused (linear1, 8 x 128x52) and
                 k_dw2_hbar (linear2 from the per-sample Hbar, 32x128) in mapper mode
  bytes per SAMPLE POINT
    sample/interpolate  8 x 128 B feature rows per decoder gathered + 128 B written per decoder + 80 B
                        neighbour list/weights/count/z  (the grid candidate scan is extra, not counted)
    feature scatter     8 x 128 B read-modify-write per decoder + the gradient rows read (k_feat_gather)
    weight gradients    5 424 B of saved rows per sample (k_wgrad: d y 640 + layer inputs 512 + e 40 + c 32 + h4 128 + d logit 4 floats)
    rel-pos backward    mapper mode: 8 x 128 B feature-row gradients + 512 B Hbar + 4 B written per sample (no neighbour rows)
"""
import ctypes as C

PEAK_F32_MFMA_TFLOPS = 157.3
# fp32 products as six bf16 piece products on the bf16 matrix pipe (lk_common.h::lk_mma6): dense bf16 peak 2516.6 TFLOP/s
# (32x32x16 at 32 cycles, 256 CUs x 4 SIMDs x 2.4 GHz) / 6 instructions per fp32-equivalent product block
PEAK_BF16_MFMA_TFLOPS = 2516.6
PEAK_F32_VIA_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# forward kernels: two fp16 pieces, three products (lk_common.h::lk_mma3h) on the same pipe (fp16 rate = bf16 rate)
PEAK_F32_VIA_F16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0
# matrix path of each MFMA-bound kernel: 'f32' = v_mfma_f32_32x32x2_f32, 'bf16x6' / 'f16x3' = split products
# (k_relpos_bwd: scaled fp16 pieces in mapper mode, bf16 pieces in tracker mode - work_per_step splits its flops; k_decode_bwd: the mapper's
# launches, k_decode_bwd_track the tracker's, whose geometry role stays on bf16 pieces;
# k_wgrad: scaled fp16 pieces in mapper mode, its only caller in the benchmark)
MFMA_PATH = {'k_decode_fwd': 'f16x3', 'k_relpos_fwd': 'f16x3', 'k_relpos_bwd': 'bf16x6', 'k_decode_bwd': 'f16x3', 'k_decode_bwd_track': 'bf16x6',
             'k_wgrad': 'f16x3'}
S = 5

MAC = dict(
    dec_fwd_geo=15200, dec_fwd_col=96640,
    dec_bwd_geo=3 * 32 * 32 + 32 * 128 + 32 * 96 + 5 * 32 * 32 + 32,
    dec_bwd_col=3 * 128 * 128 + 128 * 128 + 5 * 128 * 32 + 3 * 128, dec_bwd_track_extra=2 * 128 * 40,
    rel_fwd=8 * (128 * 52 + 32 * 128), rel_bwd=8 * (2 * 128 * 52 + 32 * 128), rel_bwd_track_extra=8 * 32 * 128,
    wgrad_col=128 * 40 + 3 * 128 * 128 + 128 * 168 + 3 * 128 + 4 * 128 * 32 + 3 * 32, rel_dw1=8 * 128 * 52,
)


class KernelTimer:
    def __init__(self, eng, names='*'):
        self.eng, self.names = eng, names

    def start(self):
        self.eng.lib.check(self.eng.lib.dll.lk_profile_begin(self.names.encode()), 'lk_profile_begin')

    def stop(self):
        buf = C.create_string_buffer(4096)
        self.eng.lib.check(self.eng.lib.dll.lk_profile_end(buf, 4096), 'lk_profile_end')
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = dict(calls=int(calls), total_ms=float(ms))
        return out


def work_per_step(b):
    """kernel -> dict(flops=, bytes=, launches=) : algorithmic work per benchmark step for budget b (either may be 0:
    a kernel is priced against the roof that takes longer, flops / MFMA peak or bytes / HBM peak)."""
    Pm, Pt = b.map_rays * S, b.track_rays * S
    n_geo, n_col, n_trk = b.map_geo_iters, b.map_iters - b.map_geo_iters, b.track_iters
    rel = 1 if b.rel_pos else 0
    fl = lambda macs: 2.0 * macs
    feat_rows = 8 * 128
    # bytes per sample read by the weight-gradient reductions: d y (640) + layer inputs h0..h3 (512) + e (40) + c (32) + h4 (128)
    # + d logit (4) floats
    wg_bytes_col = 4.0 * (640 + 512 + 40 + 32 + 128 + 4)
    w = {
        # (the tracker's launches are k_relpos_decode_fwd: the rel-pos MLP of their samples runs in the same launch, timed under this name)
        'k_decode_fwd': dict(flops=fl(n_geo * Pm * MAC['dec_fwd_geo'] + (n_col * Pm + n_trk * Pt) * (MAC['dec_fwd_geo'] + MAC['dec_fwd_col']) +
                                      rel * n_trk * Pt * MAC['rel_fwd']),
                             bytes=0.0, launches=b.map_iters + n_trk),
        # the mapper's launches (the <., ., true> instantiations: both decoder roles on scaled fp16 pieces) ...
        'k_decode_bwd': dict(flops=fl(n_geo * Pm * MAC['dec_bwd_geo'] + n_col * Pm * (MAC['dec_bwd_geo'] + MAC['dec_bwd_col'])),
                             bytes=0.0, launches=b.map_iters),
        # ... and the tracker's (other instantiations - rocprofv3 lists them as kernels of their own, and so does the timer: lumped under
        # one name their 100 launches per step traded places with k_wgrad's 36 from run to run, and events around EVERY launch of the
        # launch stream's chain cost the timed region 0.7 ms per step where events around k_wgrad's side-stream launches cost nothing):
        # colour role on fp16 pieces (unit-scale loss gradients), geometry role on bf16 pieces (d depth = 1 / sqrt(var) is unbounded)
        'k_decode_bwd_track': dict(flops=fl(n_trk * Pt * (MAC['dec_bwd_geo'] + MAC['dec_bwd_col'] + MAC['dec_bwd_track_extra'])),
                                   flops_by_path={'f16x3': fl(n_trk * Pt * (MAC['dec_bwd_col'] + MAC['dec_bwd_track_extra'])),
                                                  'bf16x6': fl(n_trk * Pt * MAC['dec_bwd_geo'])},
                                   bytes=0.0, launches=n_trk),
        'k_wgrad': dict(flops=fl(n_col * Pm * MAC['wgrad_col']), bytes=n_col * Pm * wg_bytes_col, launches=n_col),
        'k_sample_interp': dict(flops=0.0, bytes=float(n_geo * Pm * (feat_rows + 128 + 80) + n_col * Pm * ((2 - rel) * (feat_rows + 128) + 80) +
                                                       n_trk * Pt * ((2 - rel) * (feat_rows + 128) + 80)), launches=b.map_iters + n_trk),
        # feature-gradient scatter (mapper only): per table 8 x 128 B read-modify-write (+ 8 x 128 B of per-neighbour
        # gradients read in rel-pos mode, else the 128-B d c row) + 64 B of neighbour ids / weights
        'k_feat_gather': dict(flops=0.0, bytes=float(b.map_iters * Pm * (2 * feat_rows + 128 + 64) +
                                                      n_col * Pm * (2 * feat_rows + (feat_rows if rel else 128))), launches=b.map_iters),
    }
    if rel:
        w['k_relpos_fwd'] = dict(flops=fl(n_col * Pm * MAC['rel_fwd']), bytes=0.0, launches=n_col)      # mapper launches only (see k_decode_fwd)
        w['k_relpos_bwd'] = dict(flops=fl(n_col * Pm * (MAC['rel_bwd'] + MAC['rel_dw1']) + n_trk * Pt * (MAC['rel_bwd'] + MAC['rel_bwd_track_extra'])),
                                 flops_by_path={'f16x3': fl(n_col * Pm * (MAC['rel_bwd'] + MAC['rel_dw1'])),
                                                'bf16x6': fl(n_trk * Pt * (MAC['rel_bwd'] + MAC['rel_bwd_track_extra']))},
                                 # mapper: d feat [8][32] + Hbar [128] + weight sum written per sample
                                 bytes=n_col * Pm * 4.0 * (8 * 32 + 128 + 1), launches=n_col + n_trk)
    return w


def step_flops(b):
    """Algorithmic fp32 FLOPs of one step of budget b: every MLP product of the forward, the backward-data and the weight-gradient
    passes of its iterations (the sum of work_per_step's flops; interpolation, composite, losses and Adam are not counted)."""
    return sum(v['flops'] for v in work_per_step(b).values())


def pmc_traffic(kernel, path=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (the newest profiles/r<N>_rocprof_summary.md, written by
    tools/summarize_prof.py from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this bench; FETCH_SIZE doubled per
    the gfx950 note of MI355X_MICROARCH.md): (bytes, source) or (None, None).  bench.py cannot collect PMC counters live."""
    import glob
    import os
    import re
    if path is None:
        cands = glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r*_rocprof_summary.md'))
        if not cands:
            return None, None
        path = max(cands, key=lambda f: int(re.match(r'r(\d+)_', os.path.basename(f)).group(1)))
    try:
        tot_n, tot_b = 0, 0.0
        for line in open(path):
            c = [x.strip() for x in line.strip().strip('|').split('|')]
            if len(c) == 5 and c[0].split('<')[0] == kernel and c[1].isdigit():         # template variants: launch-weighted mean
                tot_n += int(c[1])
                tot_b += int(c[1]) * (float(c[3]) * 1e6 + float(c[4]) * 1024.0)
        if tot_n:
            return tot_b / tot_n, f'profiles/{os.path.basename(path)} (separate --pmc FETCH_SIZE / WRITE_SIZE passes)'
    except (OSError, ValueError):
        pass
    return None, None


# rocprofv3 kernel name (template arguments stripped) -> the timer name it is booked under (lk_api.hip: kKernelNames)
TIMER_OF = {'k_decode_fwd': 'k_decode_fwd', 'k_relpos_decode_fwd': 'k_decode_fwd', 'k_relpos_fwd': 'k_relpos_fwd', 'k_wgrad': 'k_wgrad',
            'k_feat_gather': 'k_feat_gather', 'k_relpos_bwd_fused': 'k_relpos_bwd', 'k_relpos_bwd': 'k_relpos_bwd', 'k_relpos_interp_bwd': 'k_relpos_bwd',
            'k_sample_interp': 'k_sample_interp', 'k_sample_interp_pose': 'k_sample_interp', 'k_interp_repack': 'k_sample_interp'}


def stage_traffic(path=None):
    """The committed per-STAGE counter table (profiles/r<N>_stage_traffic.json, tools/stage_traffic.sh: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes over ONE iteration type at a time): {mode: {kernels: {name: {launches_per_iteration, read_mb_per_launch, ...}}}} or None."""
    import glob
    import json
    import os
    import re
    if path is None:
        cands = glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r*_stage_traffic.json'))
        if not cands:
            return None, None
        path = max(cands, key=lambda f: int(re.match(r'r(\d+)_', os.path.basename(f)).group(1)))
    try:
        with open(path) as f:
            return json.load(f)['modes'], f'profiles/{os.path.basename(path)}'
    except (OSError, ValueError, KeyError):
        return None, None


def measured_bytes_per_step(b, kernel, modes):
    """L2-fabric bytes (read + written) of the launches booked under timer name `kernel` in one step of budget b, from the per-stage
    table: sum over the iteration types of iterations x launches per iteration x MB per launch.  The decoder backward's two names are the
    tracker's (`k_decode_bwd<.., .., false>` with ray gradients) and the mapper's launches of the same rocprofv3 kernel family."""
    n_it = {'color': b.map_iters - b.map_geo_iters, 'geo': b.map_geo_iters, 'track': b.track_iters}
    tot, seen = 0.0, False
    for mode, n in n_it.items():
        for name, k in (modes.get(mode, {}).get('kernels') or {}).items():
            base = name.split('<')[0]
            if base == 'k_decode_bwd':
                t = 'k_decode_bwd_track' if mode == 'track' else 'k_decode_bwd'
            else:
                t = TIMER_OF.get(base)
            if t != kernel:
                continue
            seen = True
            tot += n * k['launches_per_iteration'] * (k['read_mb_per_launch'] + k['written_mb_per_launch']) * 1e6
    return tot if seen else None


def dominant_kernel(kstat):
    """The kernel with the largest summed duration over the profiled step; kernels within 3 % of it are level (run-to-run scatter) and the
    longer AVERAGE LAUNCH decides.  (Rounds 4-5: the decoder backward's tracker and mapper launches were timed under ONE name and their
    100 launches per step traded places with k_wgrad's 36; they are two kernels in rocprofv3's table and two names here since.)"""
    if not kstat:
        return None
    top = max(v['total_ms'] for v in kstat.values())
    level = {k: v for k, v in kstat.items() if v['total_ms'] >= 0.97 * top}
    return max(level.items(), key=lambda kv: kv[1]['total_ms'] / max(kv[1]['calls'], 1))[0]


def roofline(kstat, budget, kernel):
    """achieved = algorithmic work of all launches of `kernel` in the timed region / their summed duration, against the
    roof (fp32 MFMA or HBM) that bounds this kernel more tightly."""
    k = kstat.get(kernel)
    model = work_per_step(budget).get(kernel)
    if not k or k['total_ms'] <= 0 or model is None:
        return None
    n_steps = k['calls'] / model['launches']
    secs = k['total_ms'] * 1e-3
    flops, nbytes = model['flops'] * n_steps, model['bytes'] * n_steps
    path = MFMA_PATH.get(kernel, 'f32')
    peaks = {'bf16x6': PEAK_F32_VIA_BF16X6_TFLOPS, 'f16x3': PEAK_F32_VIA_F16X3_TFLOPS, 'f32': PEAK_F32_MFMA_TFLOPS}
    peak_mfma = peaks.get(path, PEAK_F32_MFMA_TFLOPS)
    t_mfma = flops / (peak_mfma * 1e12)
    if model.get('flops_by_path'):        # launches on different piece types: the roof is the time-weighted blend
        t_mfma = sum(f * n_steps / (peaks[p] * 1e12) for p, f in model['flops_by_path'].items())
        peak_mfma = flops / t_mfma / 1e12 if t_mfma > 0 else peak_mfma
        path = '+'.join(sorted(p for p, f in model['flops_by_path'].items() if f > 0))
    t_hbm = nbytes / (PEAK_HBM_GBS * 1e9)
    out = {'kernel': kernel, 'launches': k['calls'], 'avg_launch_us': 1e3 * k['total_ms'] / k['calls'], 'traffic': None}
    # `achieved` is ALGORITHMIC work / time (the work model above: DESIGN.md section 3's bytes and MACs per sample x the samples of the launch).
    # Which roof binds the kernel is decided from MEASURED bytes too where the per-stage counter table has them (round-5 review: the training
    # forward was labelled `mfma` at 0.11 of the matrix roof while it WRITES 115-146 MB of saved activations per launch - it sits at 0.3-0.4
    # of the HBM roof): a kernel whose model says `mfma` but whose counter bytes / 8 TB/s exceed its matrix time is bound by the bytes it
    # really moves, and only then `achieved` is counter bytes / time (`achieved_from` says so; the model's bytes stay beside it).  For a
    # kernel the model itself puts on the HBM roof (k_wgrad) the counter bytes are reported (`measured_bytes_per_launch_avg`,
    # `frac_of_hbm_peak_measured_bytes`) and never replace the algorithmic figure.
    model_bytes = nbytes
    modes, src_st = stage_traffic()
    meas = measured_bytes_per_step(budget, kernel, modes) if modes else None
    if meas is not None:
        out['measured_bytes_per_launch_avg'] = meas / model['launches']
        out['measured_bytes_source'] = src_st
        t_meas = meas * n_steps / (PEAK_HBM_GBS * 1e9)
        out['t_mfma_over_t_hbm_measured'] = (t_mfma / t_meas) if t_meas > 0 else None
        out['frac_of_hbm_peak_measured_bytes'] = meas * n_steps / secs / 1e9 / PEAK_HBM_GBS
        if t_mfma >= t_hbm and t_meas > t_mfma:
            nbytes, t_hbm = meas * n_steps, t_meas
            out['bound_from'] = out['achieved_from'] = 'measured counter bytes (the work model has the kernel on the matrix roof)'
    if t_mfma >= t_hbm:
        out.update(bound='mfma', achieved=flops / secs / 1e12, peak=peak_mfma, unit='TFLOP/s', mfma_path=path,
                   frac_of_f32_mfma_peak=flops / secs / 1e12 / PEAK_F32_MFMA_TFLOPS)
    else:
        out.update(bound='hbm', achieved=nbytes / secs / 1e9, peak=PEAK_HBM_GBS, unit='GB/s')
    out['frac'] = out['achieved'] / out['peak']
    if kernel in ('k_wgrad', 'k_relpos_bwd', 'k_feat_gather'):
        out['overlap'] = ('k_wgrad runs on a second stream beside k_relpos_bwd / k_feat_gather: durations are measured while '
                          'they share the chip (LK_SERIAL=1 times every kernel alone)')
    out['traffic'], src = pmc_traffic(kernel)
    if src:
        out['traffic_unit'], out['traffic_source'] = 'bytes per launch (HBM read + write)', src
    out['algorithmic_flops_per_launch_avg'] = flops / k['calls']
    out['algorithmic_bytes_per_launch_avg'] = model_bytes / k['calls']
    return out
