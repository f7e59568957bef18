"""Rewrite every packed-fp32 VOP3P instruction of an AMDGPU .s file into its two scalar halves (same registers, same order of
evaluation where the halves are independent).  v_pk_{mul,add,fma}_f32 and v_pk_mov_b32 with VGPR-pair / inline-constant
operands and op_sel / op_sel_hi / neg_lo / neg_hi modifiers."""
import re, sys
src, dst = sys.argv[1], sys.argv[2]
out, n = [], 0
def half(tok, hi, neg):
    tok = tok.strip()
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m:
        r = f'v{int(m.group(1)) + (1 if hi else 0)}'
    else:
        ms = re.match(r's\[(\d+):(\d+)\]$', tok)
        r = f's{int(ms.group(1)) + (1 if hi else 0)}' if ms else tok      # SGPR pair, or an inline constant (same value for both halves)
    return ('-' + r) if neg else r
for line in open(src):
    m = re.match(r'^\t(v_pk_(mul|add|fma)_f32|v_pk_mov_b32)\s+(.*)$', line.rstrip('\n'))
    if not m:
        out.append(line)
        continue
    op, rest = m.group(2) or 'mov', m.group(3)
    mods = dict(op_sel=None, op_sel_hi=None, neg_lo=None, neg_hi=None)
    for k in list(mods):
        mm = re.search(k + r':\[([01,]+)\]', rest)
        if mm:
            mods[k] = [int(x) for x in mm.group(1).split(',')]
            rest = rest.replace(mm.group(0), '')
    ops = [t.strip() for t in rest.split(',') if t.strip()]
    d, srcs = ops[0], ops[1:]
    ns = len(srcs)
    sel = mods['op_sel'] or [0] * ns
    selh = mods['op_sel_hi'] or [1] * ns
    nlo = mods['neg_lo'] or [0] * ns
    nhi = mods['neg_hi'] or [0] * ns
    sel += [0] * (ns - len(sel)); selh += [1] * (ns - len(selh)); nlo += [0] * (ns - len(nlo)); nhi += [0] * (ns - len(nhi))
    dm = re.match(r'v\[(\d+):(\d+)\]$', d)
    d0, d1 = int(dm.group(1)), int(dm.group(2))
    lo_ops = [half(s, sel[i], nlo[i]) for i, s in enumerate(srcs)]
    hi_ops = [half(s, selh[i], nhi[i]) for i, s in enumerate(srcs)]
    if op == 'mov':          # D.lo = S0[op_sel[0]], D.hi = S1[op_sel_hi[1]]
        lo_ops, hi_ops = [half(srcs[0], sel[0], 0)], [half(srcs[1], selh[1], 0)]
    name = {'mul': 'v_mul_f32_e64', 'add': 'v_add_f32_e64', 'fma': 'v_fma_f32', 'mov': 'v_mov_b32_e32'}[op]
    lo = f'\t{name} v{d0}, ' + ', '.join(lo_ops)
    hi = f'\t{name} v{d1}, ' + ', '.join(hi_ops)
    # order: the low write must not clobber a register the high op still reads (and vice versa)
    reads_hi = set(re.findall(r'v(\d+)', ' '.join(hi_ops)))
    reads_lo = set(re.findall(r'v(\d+)', ' '.join(lo_ops)))
    if str(d0) not in reads_hi:
        out += [lo + '\n', hi + '\n']
    elif str(d1) not in reads_lo:
        out += [hi + '\n', lo + '\n']
    else:
        raise SystemExit('cross dependency: ' + line)
    n += 1
open(dst, 'w').writelines(out)
print('rewrote', n, 'packed instructions')
