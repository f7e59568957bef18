#!/usr/bin/env python3
"""Build libloopyhip.so (gfx950 code object + host stubs) in-tree with hipcc.

    python loopy_slam_amd/csrc/build.py [--force] [--verbose]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to
the GPU box with the gpurun snapshot.  Sources: every *.hip in this directory.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, 'libloopyhip.so')
OBJ = os.path.join(HERE, '_obj')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# NO PACKED FP32 (target feature packed-fp32-ops off, and no SLP vectoriser to ask for it): with v_pk_{mul,add,fma}_f32 in
# k_decode_bwd one register of the wave came out wrong in lanes 48-63 in 1-2 % of the tiles as soon as two workgroups shared a
# compute unit; the SAME instruction stream with every packed instruction rewritten into its two scalar halves is clean
# (DESIGN.md §3 "packed fp32", tools/probe/make_hist_variants.sh).  tests/test_product_hygiene.py disassembles the library
# and fails if a packed fp32 instruction is in it.  (They also take two issue slots beside the matrix instructions: 0.8 % of
# the step.)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffp-contract=off', '-fno-slp-vectorize', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(HERE, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(HERE, '*.h'))) + [os.path.join(PKG, '..', 'include', 'loopy_hip.h')]
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, 'flags.txt')
    cur = ' '.join([HIPCC] + FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != cur:
        force = True
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        if force or _newer([s] + hdrs, o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    failed = False
    with cf.ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs)))) as ex:
        for s, rc, log in ex.map(cc, jobs):
            if rc != 0 or (verbose and log.strip()):
                print(f'--- {os.path.basename(s)} (rc={rc})\n{log}', file=sys.stderr)
            failed |= rc != 0
    if failed:
        raise RuntimeError('hipcc failed')
    open(stamp, 'w').write(cur)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in srcs]
    relink = force or bool(jobs) or _newer(objs, OUT)
    print(f'loopy_slam_amd build: {len(jobs)} of {len(srcs)} sources compiled with {os.path.basename(HIPCC)} --offload-arch=gfx950 '
          f'({"relinked" if relink else "up to date"}: {os.path.relpath(OUT, os.path.dirname(PKG))})', flush=True)
    if relink:
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError('link failed')
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
