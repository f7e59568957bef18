#!/usr/bin/env python3
"""profiles/<tag>_accuracy.json: the oracle's SLAM-loop runs (fixtures, tests/golden/accuracy_<name>_oracle_s*.json) next to the product's runs
on the GPU (gpurun_out/accuracy_<name>.json written by tests/test_accuracy.py, gpurun_out/acc_<name>_hip_s*.json written by
tools/accuracy_run.py --pipeline hip) - ATE RMSE, rendered-depth L1, rotation error, the prior-only baselines.

    python tools/accuracy_summary.py r4"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r4'
KEYS = ('ate_rmse_cm', 'rot_err_deg', 'depth_l1_cm', 'max_translation_error_cm', 'wall_s', 'points')


def brief(d):
    return {k: (round(d[k], 4) if isinstance(d[k], float) else d[k]) for k in KEYS if k in d}


def stats(v):
    v = np.asarray(v, float)
    if v.size == 0:
        return None
    return {'median': round(float(np.median(v)), 4), 'mean': round(float(v.mean()), 4), 'sd': round(float(v.std(ddof=1)), 4) if v.size > 1 else None, 'min': round(float(v.min()), 4),
            'max': round(float(v.max()), 4), 'n': int(v.size)}


out = {'what': 'product (libloopyhip on one MI355X) against the CPU oracle chained into the reference\'s track + map loop (tests/oracle_slam.py) on the same '
               'synthetic hand-held sequence through the furnished room; metrics of src/tools/eval_ate.py:44-79,195-234 and src/Mapper.py:1146-1182 '
               '(depth L1 on a stride-4 pixel grid); random-init decoders on both sides; the trajectories are chaotic, the metrics are compared',
       'configs': {}}
for name in ('room', 'roomfull', 'tum', 'scannet'):
    fx = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', f'accuracy_{name}_oracle_s*.json')))]
    if not fx:
        continue
    hip = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', f'acc_{name}_hip_s*.json')))]
    test = os.path.join(ROOT, 'gpurun_out', f'accuracy_{name}.json')
    c = {'config': fx[0]['config'], 'prior_only': fx[0]['prior_only'],
         'oracle_runs': [dict(seed=o['config']['seed'], **brief(o)) for o in fx],
         'oracle': {'ate_rmse_cm': stats([o['ate_rmse_cm'] for o in fx]), 'depth_l1_cm': stats([o['depth_l1_cm'] for o in fx]),
                    'rot_err_deg': stats([o['rot_err_deg'] for o in fx])}}
    runs = [dict(seed=h['config']['seed'], **brief(h)) for h in hip]
    if os.path.exists(test):
        t = json.load(open(test))
        for k in ('ate_rmse_cm', 'depth_l1_cm'):          # the Welch records of tests/test_accuracy.py (round 6): interval, resolvable difference
            if isinstance(t.get(k), dict) and 'resolvable_rel' in t[k]:
                c.setdefault('welch', {})[k] = {q: (round(v, 5) if isinstance(v, float) else v) for q, v in t[k].items()}
        if 'statement' in t:
            c['statement'] = t['statement']
        if 'runs' in t:
            runs += [dict(seed=r['seed'], ate_rmse_cm=round(r['hip_ate'], 4), depth_l1_cm=round(r['hip_l1'], 4), rot_err_deg=round(r['hip_rot'], 4),
                          wall_s=r['hip_wall_s'], source='tests/test_accuracy.py') for r in t['runs']]
        else:       # the band form (one oracle run, three product runs)
            runs += [dict(seed=fx[0]['config']['seed'] + k, ate_rmse_cm=round(a, 4), depth_l1_cm=round(l, 4), source='tests/test_accuracy.py')
                     for k, (a, l) in enumerate(zip(t['hip_ate_rmse_cm'], t['hip_depth_l1_cm']))]
    if runs:
        c['hip_runs'] = runs
        c['hip'] = {'ate_rmse_cm': stats([r['ate_rmse_cm'] for r in runs]), 'depth_l1_cm': stats([r['depth_l1_cm'] for r in runs]),
                    'rot_err_deg': stats([r['rot_err_deg'] for r in runs if 'rot_err_deg' in r])}
        c['ate_mean_hip_over_oracle'] = round(c['hip']['ate_rmse_cm']['mean'] / c['oracle']['ate_rmse_cm']['mean'], 3)
        c['depth_l1_mean_hip_over_oracle'] = round(c['hip']['depth_l1_cm']['mean'] / c['oracle']['depth_l1_cm']['mean'], 3)
        c['ate_vs_prior'] = {'one_step_over_hip': round(c['prior_only']['one_step_ate_cm'] / c['hip']['ate_rmse_cm']['mean'], 2),
                             'dead_reckoning_over_hip': round(c['prior_only']['dead_reckoning_ate_cm'] / c['hip']['ate_rmse_cm']['mean'], 2)}
    out['configs'][name] = c
full = os.path.join(ROOT, 'gpurun_out', 'acc_room_hip_fullrays.json')
if os.path.exists(full):
    out['room_at_the_full_replica_ray_budget_hip'] = brief(json.load(open(full)))
dst = os.path.join(ROOT, 'profiles', f'{tag}_accuracy.json')
with open(dst, 'w') as f:
    json.dump(out, f, indent=1)
print(dst)
for n, c in out['configs'].items():
    if 'statement' in c:
        print(n, ':', c['statement'])
    print(n, 'oracle ATE', c['oracle']['ate_rmse_cm'], '| hip ATE', c.get('hip', {}).get('ate_rmse_cm'), '| L1 ratio', c.get('depth_l1_mean_hip_over_oracle'))
