#!/bin/bash
# Rebuild the round-1 failure ("store-data rule") from history, in the build container:
#   commit 2401a79 = first commit with the split-bf16 colour backward AND the d h register copy that hid the failure.
# Variants (each a full source tree under ab/hist_<name>/ with its own libloopyhip.so; ab/ is git-ignored but travels with gpurun):
#   copy        the commit as it is                                   -> deterministic
#   nocopy      the d h stores read the accumulators (copy removed)   -> FAILS  (wrong lanes 48-63 of one register, ~1-2 % of tiles)
#   nops_after / nops_before / vmcnt   nocopy + 64 idle cycles after / before the d h stores, + s_waitcnt vmcnt(0) after -> all FAIL
#   nocopy_noslp   nocopy built with -fno-slp-vectorize (no packed fp32)  -> deterministic
#   nocopy_pknop   nocopy, `s_nop 1` after every v_pk_* instruction (ISA edited)  -> FAILS
#   nocopy_pkscalar  nocopy, every v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 rewritten into its two scalar halves (ISA edited,
#                    tools/probe/unpack_pk.py), nothing else changed  -> deterministic
# On the GPU box:  for v in ...; do (cd ab/hist_$v && python -m pytest tests/test_fullsize_gpu.py -k deterministic_at_scale -q); done
#                  (cd ab/hist_copy && python ../../tools/probe/hist_diag.py save /tmp/ref.pt); (cd ab/hist_nocopy && python ../../tools/probe/hist_diag.py cmp /tmp/ref.pt)
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
LLVM=/opt/rocm/lib/llvm/bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
for v in copy nocopy nops_after nops_before vmcnt; do
  rm -rf ab/hist_$v; mkdir -p ab/hist_$v
  git archive 2401a79 | tar -x -C ab/hist_$v
  rm -rf ab/hist_$v/profiles ab/hist_$v/SURVEY.md
done
python3 - <<'PY'
base = 'ab/hist_%s/loopy_slam_amd/csrc/lk_bwd.hip'
src = open(base % 'copy').read()
old1 = '''        for (int q = 0; q < 16; ++q) { float t = dh[q]; asm volatile("" : "+v"(t)); dhc[q] = t; }
        if (want_w) ct_store32(a.dh_col + (size_t)sp * 640 + i * 128 + w * 32, dhc, live, lane);
'''
old2 = '''        for (int q = 0; q < 16; ++q) asm volatile("" :: "v"(dhc[q]));
'''
assert old1 in src and old2 in src
NOPS = '''        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
'''
VM = NOPS.replace('s_nop 15\\ns_nop 15\\ns_nop 15\\ns_nop 15', 's_waitcnt vmcnt(0)')
store = '''        if (want_w) ct_store32(a.dh_col + (size_t)sp * 640 + i * 128 + w * 32, dh, live, lane);
'''
for k, v in {'nocopy': store, 'nops_after': store + NOPS, 'nops_before': NOPS + store, 'vmcnt': store + VM}.items():
    s = src.replace(old1, '        for (int q = 0; q < 16; ++q) dhc[q] = dh[q];\n' + v).replace(old2, '        for (int q = 0; q < 1; ++q) (void)dhc;\n')
    open(base % k, 'w').write(s)
PY
for v in copy nocopy nops_after nops_before vmcnt; do (cd ab/hist_$v && python3 loopy_slam_amd/csrc/build.py > /dev/null) & done; wait
rm -rf ab/hist_nocopy_noslp; cp -r ab/hist_nocopy ab/hist_nocopy_noslp
sed -i "s/'-ffp-contract=off'\]/'-ffp-contract=off', '-fno-slp-vectorize']/" ab/hist_nocopy_noslp/loopy_slam_amd/csrc/build.py
(cd ab/hist_nocopy_noslp && rm -rf loopy_slam_amd/csrc/_obj loopy_slam_amd/libloopyhip.so && python3 loopy_slam_amd/csrc/build.py > /dev/null)
# ISA-edited variants: device assembly of lk_bwd.hip -> edit -> assemble -> link -> bundle -> host object with the edited code object
T=$(mktemp -d)
SRC=$ROOT/ab/hist_nocopy/loopy_slam_amd/csrc
/opt/rocm/bin/hipcc $FL --cuda-device-only -S $SRC/lk_bwd.hip -o $T/pk.s 2>/dev/null
sed 's/^\(\tv_pk_[a-z0-9_]*f32 .*\)$/\1\n\ts_nop 1/' $T/pk.s > $T/pknop.s
python3 tools/probe/unpack_pk.py $T/pk.s $T/pkscalar.s
for v in pknop pkscalar; do
  rm -rf ab/hist_nocopy_$v; cp -r ab/hist_nocopy ab/hist_nocopy_$v
  D=$ROOT/ab/hist_nocopy_$v/loopy_slam_amd
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/$v.s -o $T/$v.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/$v.out $T/$v.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/$v.out -output=$T/$v.hipfb
  /opt/rocm/bin/hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/$v.hipfb -c $D/csrc/lk_bwd.hip -o $D/csrc/_obj/lk_bwd.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libloopyhip.so $D/csrc/_obj/*.o
done
ls -la ab/hist_*/loopy_slam_amd/libloopyhip.so
