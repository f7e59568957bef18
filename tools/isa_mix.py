"""Static instruction mix of the kernels inside libloopyhip.so: per kernel (symbol matching the filter) the count of every
mnemonic class in the disassembly.  A static count, not a dynamic one - loops are counted once - but the decoder kernels are
almost fully unrolled, so the ratio of VALU to matrix instructions is what the SQ counters then show (profiles/*_sq_counters.md).

    python tools/isa_mix.py k_decode_bwd            # classes
    python tools/isa_mix.py k_decode_bwd --top 40   # individual mnemonics
    python tools/isa_mix.py k_decode_bwd --dump /tmp/x.s
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

OBJDUMP, OBJCOPY = '/opt/rocm/lib/llvm/bin/llvm-objdump', '/opt/rocm/lib/llvm/bin/llvm-objcopy'


def klass(m):
    if m.startswith('v_mfma'):
        return 'mfma'
    if m.startswith('v_cvt'):
        return 'valu:cvt'
    if m.startswith(('v_perm', 'v_pack', 'v_and', 'v_or', 'v_lshl', 'v_lshr', 'v_bfi', 'v_bfe', 'v_alignb', 'v_xor')):
        return 'valu:bits'
    if m.startswith(('v_exp', 'v_log', 'v_rcp', 'v_sin', 'v_cos', 'v_sqrt', 'v_rsq', 'v_fract', 'v_rndne', 'v_floor')):
        return 'valu:trans'
    if m.startswith(('v_mov', 'v_accvgpr', 'v_cndmask', 'v_readlane', 'v_readfirstlane', 'v_writelane', 'v_swap')):
        return 'valu:move'
    if m.startswith(('v_cmp', 'v_max', 'v_min', 'v_med')):
        return 'valu:cmp'
    if m.startswith('v_'):
        return 'valu:arith'
    if m.startswith('ds_'):
        return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem:' + ('scratch' if m.startswith('scratch') else ('store' if 'store' in m else ('atomic' if 'atomic' in m else 'load')))
    if m.startswith('s_waitcnt'):
        return 's:wait'
    if m.startswith('s_nop'):
        return 's:nop'
    if m.startswith('s_barrier'):
        return 's:barrier'
    if m.startswith('s_load') or m.startswith('s_buffer'):
        return 's:load'
    return 's:other'


def kernels(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([OBJCOPY, '--dump-section=.hip_fatbin=' + fat, lib], check=True)
        blob = open(fat, 'rb').read()
        offs = [m.start() for m in re.finditer(b'\x7fELF', blob)]
        for k, o in enumerate(offs):
            co = os.path.join(tmp, 'co.elf')
            open(co, 'wb').write(blob[o:offs[k + 1] if k + 1 < len(offs) else len(blob)])
            dis = subprocess.run([OBJDUMP, '-d', '-C', co], capture_output=True, text=True).stdout
            name, body = None, []
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    if name:
                        yield name, body
                    name, body = m.group(1), []
                elif name and line.startswith('\t'):
                    body.append(line.strip())
            if name:
                yield name, body


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('filter')
    ap.add_argument('--top', type=int, default=0)
    ap.add_argument('--dump')
    ap.add_argument('--lib', default=None)
    a = ap.parse_args()
    if a.lib is None:
        from loopy_slam_amd.csrc import build
        a.lib = build.build()
    for name, body in kernels(a.lib):
        if a.filter not in name:
            continue
        mn = [ln.split()[0] for ln in body if ln and not ln.startswith('//')]
        c = collections.Counter(klass(m) for m in mn)
        valu = sum(v for k, v in c.items() if k.startswith('valu'))
        print(f'== {name[:110]}\n   {len(mn)} instructions, valu {valu}, mfma {c["mfma"]}, valu/mfma {valu / max(1, c["mfma"]):.1f}')
        print('   ' + '  '.join(f'{k} {v}' for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        if a.top:
            cc = collections.Counter(mn)
            print('   ' + '  '.join(f'{k} {v}' for k, v in cc.most_common(a.top)))
        if a.dump:
            open(a.dump, 'a').write(f'// {name}\n' + '\n'.join(body) + '\n')
