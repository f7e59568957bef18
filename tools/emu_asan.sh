#!/bin/bash
# tools/emu_asan.sh [pytest args...]: the CPU test-suite with the host emulation of the .hip sources built under AddressSanitizer -
# out-of-bounds reads / writes of the kernels on the (ASan-allocated) torch CPU tensors and on the emulator's own allocations abort the run.
# Needs the clang of the ROCm install (its ASan runtime is preloaded into python).
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/emu_asan; mkdir -p $OUT
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
FL="-x c++ -O1 -g -std=c++17 -fPIC -mfma -mavx2 -ffp-contract=off -Itests/hipemu -Wno-unknown-pragmas -Wno-unused-value -Wno-pass-failed -fsanitize=address -fno-omit-frame-pointer -shared-libasan"
pids=()
for s in loopy_slam_amd/csrc/*.hip tests/hipemu/hipemu_runtime.cpp; do
  o=$OUT/$(basename ${s%.*}).o
  if [ ! -f $o ] || [ $s -nt $o ] || [ -n "$(find loopy_slam_amd/csrc tests/hipemu/hip include -name '*.h' -newer $o | head -1)" ]; then
    $CXX $FL -c $s -o $o & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -fPIC -fsanitize=address -shared-libasan -o $OUT/libloopyhip_emu.so $OUT/*.o
LOOPY_EMU_LIB=$OUT/libloopyhip_emu.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest "${@:-tests}" -x -q -m "not gpu"
