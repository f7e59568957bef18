"""bench.py as the driver launches it (`-m gpu`): the default single-GPU line with the three budgets, the multi-rank command line
(`python bench.py --gpus 2`: self-spawned ranks, parallel.DistContext, the phase-split loop around the exchange) on ONE device over gloo,
and one rank over RCCL with the direct communicator and with the overlapped row exchange.  No scaling figure comes out of this - RCCL
refuses two ranks on one device, the 2-rank leg stages its collectives through the host - what is pinned is that the exact command of the
driver's 8-GPU run starts, agrees on its accounting and prints ONE JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                   # ONE JSON line, the last thing on stdout
    assert r.stdout.strip().splitlines()[-1] == lines[0]
    return json.loads(lines[0])


@pytest.fixture(scope='module')
def plain():
    return _bench(['--steps', '5', '--warmup', '1', '--no-cpu-baseline'])


def test_default_line_carries_the_three_budgets(plain):
    d = plain
    assert d['n_gpus'] == 1 and d['unit'] == 'rays/s' and d['higher_is_better'] and d['value'] > 0
    assert d['config']['rays_per_step'] == 40 * 1500 + 60 * 5000
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and 0 < r['frac'] < 1 and 0 < r['whole_step_fp32_frac'] < 1
    w = d['workloads']
    assert set(w) == {'tum', 'scannet', 'replica_2m', 'refine_5m_f16'}
    assert w['tum']['rays_per_step'] == 200 * 5000 + 150 * 10000 and w['scannet']['rays_per_step'] == 100 * 5000 + 60 * 10000
    for k in ('tum', 'scannet'):
        assert w[k]['ms_per_step'] > 0 and abs(w[k]['rays_per_s'] * w[k]['ms_per_step'] * 1e-3 / w[k]['rays_per_step'] - 1) < 1e-6
        assert 0 < w[k]['whole_step_fp32_frac'] < 1
    # a TUM frame is 2.5 M rays against Replica's 0.36 M: its step is longer, its ray rate of the same order
    assert w['tum']['ms_per_step'] > 2 * d['ms_per_step'] and 0.3 < w['tum']['rays_per_s'] / d['value'] < 3
    # BASELINE configs 4 / 5 at their map sizes: the headline budget on a 2 M-point map costs what it costs at 100 k points (the search walks
    # cells, not the map: within 15 %), and the whole-map refinement call at 5 M points with half tables steps only touched rows
    big, ref = w['replica_2m'], w['refine_5m_f16']
    assert 1_990_000 < big['n_points'] < 2_010_000 and big['rays_per_step'] == d['config']['rays_per_step']
    assert 0.85 < big['ms_per_step'] / d['ms_per_step'] < 1.25
    assert 4_990_000 < ref['n_points'] < 5_010_000 and ref['iterations'] == 300 and 'float16' in ref['feature_tables']
    assert 0.05 < ref['ms_per_iteration'] < 1.0          # (dense Adam over 5 M rows was 1.0 / 2.0 ms per geometry / colour iteration, profiles/r3_refine.md)


def test_two_ranks_on_one_device_command_line():
    d = _bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
               env={'LOOPY_DIST_ONE_DEVICE': '1', 'LOOPY_DIST_BACKEND': 'gloo'})
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 2
    assert d['config']['rays_per_step_all_ranks'] == 2 * 60 * 5000 + 40 * 1500          # every rank its own mapping rays, tracking replicated: once
    assert d['value'] > 0 and d['value'] == pytest.approx(d['config']['rays_per_step_all_ranks'] * d['steps'] / (d['ms_per_step'] * 1e-3 * d['steps']), rel=1e-6)
    assert 'dp2' in d['config']['parallelism'] and 'workloads' not in d


def test_one_rank_over_rccl_exchange_overhead(plain):
    """LOOPY_DIST_FORCE=1: the data-parallel code path (phase-split loop, bucket pack, a real all-reduce that is a copy, replicated tracking's
    broadcast) with ONE rank over RCCL - what the exchange machinery costs before a second GPU is involved.  Measured (round 5, one box,
    gpurun_out/bench_dist_one_rank.json): plain loop 18.55 ms, direct communicator on the launch stream 19.56 (+1.0: per mapping iteration the
    pack launch, the one-rank RCCL kernel and the Adam launch that otherwise rides in the reduction launch - and the phase-split loop cannot
    take the split step of the plain one), overlapped row exchange 20.88 (+2.3: a second collective and two stream hand-overs per iteration
    with nothing on the wire to hide - it is the default only where ranks exchange over xGMI).  Bounds: +8 % / +16 % of the plain step (measured
    +5.4 % / +12.5 %; relative, because one box of the pool runs every kernel 1.4x slower) - a regression of a launch per iteration (0.3-0.4 ms
    per step = 2 %) fails; the round-4 review's +1.0 ms target for the direct path is on the line (1.0 measured)."""
    direct = _bench(['--steps', '5', '--warmup', '1', '--no-cpu-baseline'], env={'LOOPY_DIST_FORCE': '1', 'LOOPY_DIST_OVERLAP': '0'})
    over = _bench(['--steps', '5', '--warmup', '1', '--no-cpu-baseline'], env={'LOOPY_DIST_FORCE': '1', 'LOOPY_DIST_OVERLAP': '1'})
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'bench_dist_one_rank.json'), 'w') as f:
            json.dump({'plain_ms': plain['ms_per_step'], 'direct_ms': direct['ms_per_step'], 'overlap_ms': over['ms_per_step'],
                       'plain_iterations_ms': plain['ms_per_step_iterations'], 'direct_iterations_ms': direct['ms_per_step_iterations'],
                       'overlap_iterations_ms': over['ms_per_step_iterations']}, f, indent=1)
    assert direct['n_gpus'] == 1 and direct['ms_per_step'] <= 1.08 * plain['ms_per_step'], (plain['ms_per_step'], direct['ms_per_step'])
    assert over['ms_per_step'] <= 1.16 * plain['ms_per_step'], (plain['ms_per_step'], over['ms_per_step'])
