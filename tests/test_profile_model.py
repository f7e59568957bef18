"""The roofline bookkeeping of bench.py (loopy_slam_amd/profile.py) on the CPU: the work model per timer name adds up to the step's algorithmic
FLOPs, the decoder backward's two names (mapper / tracker launches - two kernels in rocprofv3's table) carry the launches and the work of their
own iterations only, and the dominant-kernel rule is stable against the run-to-run scatter that once flipped it."""
import pytest

from loopy_slam_amd import profile, workload


@pytest.mark.parametrize('mk', [workload.Budget, workload.Budget.tum, workload.Budget.scannet])
def test_work_model_adds_up(mk):
    b = mk()
    w = profile.work_per_step(b)
    assert profile.step_flops(b) == pytest.approx(sum(v['flops'] for v in w.values()))
    Pm, Pt = b.map_rays * profile.S, b.track_rays * profile.S
    n_col = b.map_iters - b.map_geo_iters
    M = profile.MAC
    # decoder backward: mapper launches and tracker launches under their own names, nothing counted twice
    assert w['k_decode_bwd']['launches'] == b.map_iters and w['k_decode_bwd_track']['launches'] == b.track_iters
    assert w['k_decode_bwd']['flops'] == pytest.approx(2.0 * (b.map_geo_iters * Pm * M['dec_bwd_geo'] + n_col * Pm * (M['dec_bwd_geo'] + M['dec_bwd_col'])))
    trk = w['k_decode_bwd_track']
    assert trk['flops'] == pytest.approx(2.0 * b.track_iters * Pt * (M['dec_bwd_geo'] + M['dec_bwd_col'] + M['dec_bwd_track_extra']))
    assert sum(trk['flops_by_path'].values()) == pytest.approx(trk['flops'])
    for name, v in w.items():
        assert v['launches'] > 0 and (v['flops'] > 0 or v['bytes'] > 0), name
        if v.get('flops_by_path'):
            assert sum(v['flops_by_path'].values()) == pytest.approx(v['flops']), name
        assert name in profile.MFMA_PATH or v['flops'] == 0, name


def test_dominant_kernel_rule():
    # the bench step as the timer reports it (profiles/r5_bench.json, kernel_ms_per_step): k_wgrad leads the mapper's decoder backward by a wide
    # margin once the tracker's launches have their own name ...
    k = {'k_wgrad': dict(calls=36, total_ms=3.99), 'k_relpos_bwd': dict(calls=76, total_ms=3.70), 'k_decode_fwd': dict(calls=100, total_ms=3.59),
         'k_decode_bwd': dict(calls=60, total_ms=2.72), 'k_decode_bwd_track': dict(calls=40, total_ms=1.23)}
    assert profile.dominant_kernel(k) == 'k_wgrad'
    # ... and within 3 % the longer average launch decides, whichever way the scatter went
    for a, b in ((3.99, 4.05), (4.05, 3.99)):
        k2 = dict(k, k_wgrad=dict(calls=36, total_ms=a), k_relpos_bwd=dict(calls=76, total_ms=b))
        assert profile.dominant_kernel(k2) == 'k_wgrad'
    assert profile.dominant_kernel({}) is None


def test_roofline_prices_each_form_on_its_own_pieces():
    b = workload.Budget()
    w = profile.work_per_step(b)
    # one second of each: the mapper's form on fp16 pieces (three products per fp32 product), the tracker's geometry role on bf16 pieces (six)
    for name in ('k_decode_bwd', 'k_decode_bwd_track'):
        r = profile.roofline({name: dict(calls=w[name]['launches'], total_ms=1000.0)}, b, name)
        assert r is not None and r['kernel'] == name and 0 < r['frac'] < 1
    rm = profile.roofline({'k_decode_bwd': dict(calls=w['k_decode_bwd']['launches'], total_ms=1.0)}, b, 'k_decode_bwd')
    assert rm['peak'] == pytest.approx(profile.PEAK_F32_VIA_F16X3_TFLOPS, rel=1e-6) or rm['bound'] == 'hbm'


def test_headline_roofline_is_priced_on_the_work_models_bytes():
    """`achieved` of a kernel the work model itself puts on the HBM roof (k_wgrad) is ALGORITHMIC bytes / time - 5 424 B per sample x 25 000
    samples per launch - whatever the committed counter table says; the counter bytes are reported beside it and only
    decide the LABEL of a kernel the model has on the matrix roof (the training forward, bound by the activations it stores)."""
    b = workload.Budget()
    w = profile.work_per_step(b)['k_wgrad']
    secs = 36 * 95e-6
    r = profile.roofline({'k_wgrad': dict(calls=w['launches'], total_ms=secs * 1e3)}, b, 'k_wgrad')
    per_launch = w['bytes'] / w['launches']
    assert per_launch == pytest.approx(5424 * 25000, rel=0.02)
    assert r['bound'] == 'hbm' and 'achieved_from' not in r
    assert r['algorithmic_bytes_per_launch_avg'] == pytest.approx(per_launch)
    assert r['achieved'] == pytest.approx(w['bytes'] / secs / 1e9) and r['frac'] == pytest.approx(r['achieved'] / profile.PEAK_HBM_GBS)
    if 'measured_bytes_per_launch_avg' in r:          # (the committed per-stage table: what the kernel really moves, beside the model)
        assert r['frac_of_hbm_peak_measured_bytes'] == pytest.approx(r['measured_bytes_per_launch_avg'] * w['launches'] / secs / 1e9 / profile.PEAK_HBM_GBS)
    f = profile.roofline({'k_decode_fwd': dict(calls=profile.work_per_step(b)['k_decode_fwd']['launches'], total_ms=3.5)}, b, 'k_decode_fwd')
    if f.get('achieved_from'):
        assert f['bound'] == 'hbm' and f['algorithmic_bytes_per_launch_avg'] < f['measured_bytes_per_launch_avg']


def test_work_model_against_the_decoders_own_shapes_and_the_counter_table():
    """The MACs per sample the roofline is priced on, derived independently from the shapes of the weight tensors the kernels consume
    (synthetic.default_weights: the reference's NICER decoders, decoder.py:12-43,180-288,431-626), and the model's bytes for the dominant
    kernel held against what the PMC counters of the committed per-stage table say it really moves (round-5 review: the model was checked
    only against itself)."""
    from loopy_slam_amd import synthetic as syn
    W = syn.default_weights(rel_pos=True)
    mm = lambda prefix: sum(v.shape[0] * v.shape[1] for k, v in W.items() if k.startswith(prefix) and k.endswith('.weight'))
    M = profile.MAC
    col = mm('color_decoder.pts_linears') + mm('color_decoder.fc_c') + mm('color_decoder.output_linear')
    geo = mm('geo_decoder.pts_linears') + mm('geo_decoder.fc_c') + mm('geo_decoder.output_linear')
    assert M['dec_fwd_col'] == col == 96640
    assert M['dec_fwd_geo'] == geo == 15200                                     # (+ 3 x 93 for the Fourier argument, on the vector unit)
    assert M['rel_fwd'] == 8 * mm('color_decoder.mlp_col_neighbor') == 86016    # eight neighbours per sample
    # weight gradients: one outer product per trunk / output matrix entry + the 32 auxiliary columns M_i = d y_i^T c of the four trunk jobs and
    # the output job that the fc_c gradients are formed from (k_wgrad (b))
    assert M['wgrad_col'] == mm('color_decoder.pts_linears') + mm('color_decoder.output_linear') + 4 * 128 * 32 + 3 * 32
    # backward-data: every trunk matrix transposed, minus the first layer's embedding columns (no gradient wanted there in mapper mode)
    assert M['dec_bwd_col'] == 3 * 128 * 128 + 128 * 128 + mm('color_decoder.fc_c') + mm('color_decoder.output_linear')
    # bytes of the dominant kernel: the model's rows (d y 640 + layer inputs 512 + e 40 + c 32 + h_4 128 + d logit 4 floats per sample) against the counters
    b = workload.Budget()
    w = profile.work_per_step(b)['k_wgrad']
    per_launch = w['bytes'] / w['launches']
    assert per_launch == pytest.approx(4.0 * (640 + 512 + 40 + 32 + 128 + 4) * b.map_rays * profile.S)
    modes, src = profile.stage_traffic()
    if modes:
        k = [v for n, v in modes['color']['kernels'].items() if n.startswith('k_wgrad')][0]
        read_mb = k['read_mb_per_launch']
        assert 0.85 <= per_launch / 1e6 / read_mb <= 1.02, (per_launch, read_mb, src)      # the rows are fetched 1.0-1.15 x (units that share rows, XCD by XCD)
