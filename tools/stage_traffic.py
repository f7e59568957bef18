#!/usr/bin/env python3
"""Per-STAGE traffic table from the passes of tools/stage_traffic.sh:

    python tools/stage_traffic.py TAG ITERS CALLS mode [mode ...]      # reads /tmp/st_<mode>_{time,fetch,write}/

One table per iteration type ('color', 'geo', 'track'): for every kernel its launches per iteration, average duration, and the L2's
memory-side read / write bytes per launch and per iteration, then the iteration's sum against SURVEY.md §8(d)'s algorithmic bytes.
Counter handling as /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are KiB per dispatch from the L2's fabric-side
request counters (Infinity-Cache hits are counted, not excluded: an upper bound of the HBM bytes); on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads, so the read column doubles it (an upper bound again for narrow gathers); WRITE_SIZE is quoted as is."""
import collections
import csv
import glob
import os
import sys

tag, iters, calls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
modes = sys.argv[4:] or ['color']
# forward 11.1 KB / ray, backward + 20.5 KB / ray (both tables, S = 5, k = 8); 'geometry' touches one table
ALGO = {'color': (5000, 31.6e3), 'geo': (5000, 0.5 * 31.6e3), 'track': (1500, 11.1e3)}


# kernels of the workload's SET-UP (synthetic scene, map insertion, index build - once per process, torch's own element-wise kernels): not
# part of an iteration; listed in one line under the table
SETUP = {'k_aabb', 'k_count', 'k_scatter', 'k_zero_counts', 'k_grid_finalize', 'k_add_test', 'k_add_emit', 'k_compact', 'k_frustum_sample', 'k_frustum_mask',
         'k_compact_count', 'k_compact_scan', 'k_compact_write'}


def short(n):
    return n.split('(')[0].replace('void ', '').strip()


def is_setup(k):
    return k.startswith('at::') or k in SETUP


def counter(mode, pass_, name):
    a = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f'/tmp/st_{mode}_{pass_}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                k = short(r['Kernel_Name'])
                a[k][0] += 1
                a[k][1] += float(r['Counter_Value'])
    return a


def durations(mode):
    a = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f'/tmp/st_{mode}_time/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            a[k][0] += 1
            a[k][1] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
    return a


import json
JS = {'what': 'per iteration TYPE and kernel: launches per iteration, average duration (us), L2-fabric read / written MB per launch (read = 2 x FETCH_SIZE, '
              'MI355X_MICROARCH.md §HBM); tools/stage_traffic.py', 'modes': {}}
print(f'# HBM-side traffic per iteration TYPE ({tag}): rocprofv3 passes over `python tools/mode_trace.py <mode> {iters} --repeat {calls - 1}`')
print()
print('Three passes per mode (`--kernel-trace --stats`, `--kernel-trace --pmc FETCH_SIZE`, `--kernel-trace --pmc WRITE_SIZE`); counters = the L2\'s')
print('fabric-side requests (Infinity-Cache hits included), KiB per dispatch; "read" doubles FETCH_SIZE (gfx950 tallies 128-B requests at 64 B,')
print('MI355X_MICROARCH.md §HBM); "per iteration" = sum over the run / iterations run (%d calls x %d), so launches that happen once per call or' % (calls, iters))
print('once per look-ahead chunk (batch assembly, neighbour search, row sort) are spread over the iterations they serve.')
for mode in modes:
    n_it = iters * calls
    fe, wr, du = counter(mode, 'fetch', 'FETCH_SIZE'), counter(mode, 'write', 'WRITE_SIZE'), durations(mode)
    rays, per_ray = ALGO[mode]
    rows, tot_r, tot_w, tot_us, setup_mb = [], 0.0, 0.0, 0.0, 0.0
    for k in sorted(set(fe) | set(wr) | set(du), key=lambda k: -(2 * fe[k][1] + wr[k][1])):
        n = max(fe[k][0], wr[k][0], du[k][0])
        if n == 0:
            continue
        r_mb = 2 * fe[k][1] * 1024 / 1e6
        w_mb = wr[k][1] * 1024 / 1e6
        us = du[k][1]
        if is_setup(k):
            setup_mb += r_mb + w_mb
            continue
        tot_r += r_mb; tot_w += w_mb; tot_us += us
        if (r_mb + w_mb) / n_it < 0.05 and us / n_it < 0.5:
            continue
        rows.append((k, n / n_it, us / max(du[k][0], 1), r_mb / max(fe[k][0], 1), w_mb / max(wr[k][0], 1), r_mb / n_it, w_mb / n_it))
        JS['modes'].setdefault(mode, {'rays': rays, 'kernels': {}})['kernels'][k] = dict(
            launches_per_iteration=round(n / n_it, 4), avg_us=round(us / max(du[k][0], 1), 2), read_mb_per_launch=round(r_mb / max(fe[k][0], 1), 3),
            written_mb_per_launch=round(w_mb / max(wr[k][0], 1), 3))
    print()
    print(f'## {mode} (R = {rays})')
    print()
    print('| kernel | launches / iteration | avg us | read MB / launch | written MB / launch | read MB / iteration | written MB / iteration |')
    print('|---|---|---|---|---|---|---|')
    for k, lpi, us, r1, w1, ri, wi in rows:
        print(f'| {k} | {lpi:.2f} | {us:.1f} | {r1:.1f} | {w1:.1f} | {ri:.1f} | {wi:.1f} |')
    algo = rays * per_ray / 1e6
    tot = (tot_r + tot_w) / n_it
    print()
    print(f'(left out: the set-up of the process - synthetic scene, insertion, index build, torch element-wise kernels - {setup_mb:.0f} MB in all.)')
    print()
    print(f'**Per iteration: read {tot_r / n_it:.0f} MB + written {tot_w / n_it:.0f} MB = {tot:.0f} MB; sum of kernel durations {tot_us / n_it:.0f} us; '
          f'SURVEY §8(d) algorithmic bytes {algo:.0f} MB ({rays} rays x {per_ray / 1e3:.1f} KB) -> traffic ratio {tot / algo:.2f}.**')
    JS['modes'][mode].update(read_mb_per_iteration=round(tot_r / n_it, 1), written_mb_per_iteration=round(tot_w / n_it, 1), algorithmic_mb_per_iteration=round(algo, 1),
                             traffic_ratio=round(tot / algo, 3), kernel_us_per_iteration=round(tot_us / n_it, 1))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.isdir(os.path.join(root, 'gpurun_out')):
    with open(os.path.join(root, 'gpurun_out', f'stage_traffic_{tag}.json'), 'w') as f:
        json.dump(JS, f, indent=1)
