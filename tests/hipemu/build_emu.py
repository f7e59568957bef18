#!/usr/bin/env python3
"""TEST-ONLY: compile the unmodified loopy_slam_amd/csrc/*.hip for the HOST against the HIP
emulation header in this directory -> tests/hipemu/_build/libloopyhip_emu.so.

Used by the CPU test-suite to exercise the kernels' logic without a GPU.  Never loaded by the
product package (loopy_slam_amd._ffi only ever opens libloopyhip.so).
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'loopy_slam_amd', 'csrc')
BUILD = os.path.join(HERE, '_build')
OUT = os.path.join(BUILD, 'libloopyhip_emu.so')
CXX = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
FLAGS = ['-x', 'c++', '-O2', '-std=c++17', '-fPIC', '-mfma', '-mavx2', '-ffp-contract=off',
         '-I', HERE, '-Wno-unknown-pragmas', '-Wno-unused-value', '-Wno-pass-failed']


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip'))) + [os.path.join(HERE, 'hipemu_runtime.cpp')]
    deps = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, 'hip', 'hip_runtime.h'),
                                                          os.path.join(ROOT, 'include', 'loopy_hip.h')]
    jobs = []
    for s in srcs:
        o = os.path.join(BUILD, os.path.basename(s).rsplit('.', 1)[0] + '.o')
        if force or not os.path.exists(o) or any(os.path.getmtime(x) > os.path.getmtime(o) for x in [s] + deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [CXX] + FLAGS + ['-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    failed = False
    with cf.ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs)))) as ex:
        for s, rc, log in ex.map(cc, jobs):
            if rc != 0 or (verbose and log.strip()):
                print(f'--- {os.path.basename(s)} (rc={rc})\n{log}', file=sys.stderr)
            failed |= rc != 0
    if failed:
        raise RuntimeError('emu compile failed')
    objs = [os.path.join(BUILD, os.path.basename(s).rsplit('.', 1)[0] + '.o') for s in srcs]
    if force or jobs or not os.path.exists(OUT):
        r = subprocess.run([CXX, '-shared', '-fPIC', '-o', OUT] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError('emu link failed')
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
