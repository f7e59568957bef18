#!/bin/bash
# tools/ab_build.sh NAME [extra hipcc flags...]  ->  ab/lib_NAME.so  (variant build for same-box A/B timing)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab /tmp/ab_obj_$name
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude"
pids=()
for s in loopy_slam_amd/csrc/*.hip; do
  o=/tmp/ab_obj_$name/$(basename ${s%.hip}).o
  /opt/rocm/bin/hipcc $FL "$@" -c $s -o $o 2>/tmp/ab_obj_$name/$(basename $s).log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; grep -h "error" /tmp/ab_obj_$name/*.log | head; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$name.so /tmp/ab_obj_$name/*.o
echo "ab/lib_$name.so"
