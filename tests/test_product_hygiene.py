"""Product-path rules: no oracle / emulator / CPU fallback inside loopy_slam_amd, the gfx950 library exports every
symbol include/loopy_hip.h declares, and the package fails loudly without a GPU."""
import ctypes
import glob
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_package_never_touches_oracle_or_emulator():
    bad = []
    for path in glob.glob(os.path.join(ROOT, 'loopy_slam_amd', '**', '*'), recursive=True):
        if not path.endswith(('.py', '.hip', '.h', '.cpp')):
            continue
        src = open(path).read()
        for pat in (r'^\s*(from|import)\s+oracle', r'hipemu', r'libloopyhip_emu', r'HIPEMU'):
            if re.search(pat, src, re.M):
                bad.append((os.path.relpath(path, ROOT), pat))
    # the entry points at the repository root and the `src.*` import aliases are product too
    for path in [os.path.join(ROOT, 'run.py')] + glob.glob(os.path.join(ROOT, 'src', '**', '*.py'), recursive=True):
        src = open(path).read()
        for pat in (r'^\s*(from|import)\s+oracle', r'hipemu', r'libloopyhip_emu', r'oracle_slam', r'sys\.path.*tests'):
            if re.search(pat, src, re.M):
                bad.append((os.path.relpath(path, ROOT), pat))
    assert not bad, bad
    for name in ('bench.py', '__graft_entry__.py'):
        src = open(os.path.join(ROOT, name)).read()
        assert 'libloopyhip_emu' not in src


def test_library_exports_every_declared_symbol():
    lib_path = os.path.join(ROOT, 'loopy_slam_amd', 'libloopyhip.so')
    from loopy_slam_amd.csrc import build
    build.build()                                # incremental: a no-op when the library is newer than its sources
    import torch  # noqa: F401  (torch's HIP runtime first, see loopy_slam_amd/_ffi.py)
    dll = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, 'include', 'loopy_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    names = sorted(set(re.findall(r'\b(lk_[a-z0-9_]+)\s*\(', header)))
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    dll.lk_version.restype = ctypes.c_int
    assert dll.lk_version() == 1


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_fails_loudly_without_gpu():
    from loopy_slam_amd import _ffi, core
    with pytest.raises(_ffi.LoopyError):
        _ffi.get_lib()
    with pytest.raises(_ffi.LoopyError):
        core.Engine()


def test_missing_library_message():
    from loopy_slam_amd import _ffi
    with pytest.raises(_ffi.LoopyError, match='no CPU fallback'):
        _ffi.LoopyLib('/nonexistent/libloopyhip.so')


def test_library_has_no_packed_fp32_instructions():
    """gfx950's packed fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32) corrupted lanes
    48-63 of a register in k_decode_bwd when two workgroups shared a compute unit (reproduced from history by
    tools/probe/make_hist_variants.sh).  The library is built with the packed-fp32-ops target feature off; this test
    disassembles every code object inside libloopyhip.so and fails if one slipped in (new flags, a new compiler)."""
    import subprocess
    objdump, objcopy = '/opt/rocm/lib/llvm/bin/llvm-objdump', '/opt/rocm/lib/llvm/bin/llvm-objcopy'
    if not (os.path.exists(objdump) and os.path.exists(objcopy)):
        pytest.skip('ROCm LLVM binutils not installed')
    from loopy_slam_amd.csrc import build
    lib_path = build.build()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([objcopy, '--dump-section=.hip_fatbin=' + fat, lib_path], check=True)
        blob = open(fat, 'rb').read()
        offs = [m.start() for m in re.finditer(b'\x7fELF', blob)]
        assert len(offs) >= 6                        # one code object per .hip translation unit
        found, n_insts = [], 0
        for k, o in enumerate(offs):
            co = os.path.join(tmp, 'co.elf')
            open(co, 'wb').write(blob[o:offs[k + 1] if k + 1 < len(offs) else len(blob)])
            dis = subprocess.run([objdump, '-d', co], capture_output=True, text=True).stdout
            n_insts += len(re.findall(r'\bv_mfma_f32_32x32x16_(f16|bf16)\b', dis))
            found += re.findall(r'\bv_pk_(?:mul|add|fma)_f32\b|\bv_pk_mov_b32\b', dis)
        assert n_insts > 100                         # the disassembly really is the kernels
        assert not found, f'{len(found)} packed fp32 instructions in libloopyhip.so: {sorted(set(found))}'


def _code_object_notes():
    """[(kernel name, {vgpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size})] of every kernel in libloopyhip.so
    (the AMDGPU metadata notes of its code objects)."""
    import subprocess
    import tempfile
    objcopy, readelf = '/opt/rocm/lib/llvm/bin/llvm-objcopy', '/opt/rocm/lib/llvm/bin/llvm-readelf'
    if not (os.path.exists(objcopy) and os.path.exists(readelf)):
        pytest.skip('ROCm LLVM binutils not installed')
    from loopy_slam_amd.csrc import build
    lib_path = build.build()
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([objcopy, '--dump-section=.hip_fatbin=' + fat, lib_path], check=True)
        blob = open(fat, 'rb').read()
        offs = [m.start() for m in re.finditer(b'\x7fELF', blob)]
        for k, o in enumerate(offs):
            co = os.path.join(tmp, 'co.elf')
            open(co, 'wb').write(blob[o:offs[k + 1] if k + 1 < len(offs) else len(blob)])
            notes = subprocess.run([readelf, '--notes', co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)', line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key in ('vgpr_count', 'vgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size'):
                    cur[key] = int(val)
                elif key == 'name' and val.startswith('_Z') and not val.endswith('.kd'):
                    cur['name'] = val
                elif key == 'wavefront_size':           # last key of a kernel's entry
                    if 'name' in cur:
                        out.append((cur.pop('name'), cur))
                    cur = {}
    return out


def test_mlp_kernels_keep_their_register_budget():
    """The decoder kernels run three workgroups per compute unit only below 168 registers per lane, and a spill in any of the per-sample
    kernels is a performance bug (round 3: the compiler had sunk the geometry decoder's d p chain to the end of k_decode_bwd and kept 190
    values alive for it - 209 registers, two workgroups per unit).  Only k_pregather (one launch per frame, 1 024-thread workgroups =
    128 registers) is allowed to spill."""
    notes = _code_object_notes()
    assert len(notes) > 60
    by = {}
    for name, d in notes:
        by.setdefault(name, d)
    spilled = sorted(n for n, d in by.items() if d.get('vgpr_spill_count', 0) > 0 and 'k_pregather' not in n)
    assert not spilled, spilled
    budget = {'_Z12k_decode_bwdILb1ELb0ELb1EEv15LkDecodeBwdArgsi': 168,     # mapper form: three workgroups per unit
              '_Z12k_decode_fwdILb0ELb0EEv12LkDecodeArgsi': 168,
              '_Z12k_relpos_fwd12LkRelposArgs': 168,
              '_Z18k_relpos_bwd_fusedILb1ELb0EEv15LkRelposBwdArgs': 256,     # two per unit (LDS-bound anyway)
              '_Z7k_wgradILb1EEv11LkWgradArgs': 256}
    for name, cap in budget.items():
        assert name in by, name
        assert by[name]['vgpr_count'] <= cap, (name, by[name])
    # LDS: three decoder workgroups must fit the 160 KB of a compute unit
    assert 3 * by['_Z12k_decode_bwdILb1ELb0ELb1EEv15LkDecodeBwdArgsi']['group_segment_fixed_size'] <= 160 * 1024
    assert 3 * by['_Z12k_decode_fwdILb0ELb0EEv12LkDecodeArgsi']['group_segment_fixed_size'] <= 160 * 1024
