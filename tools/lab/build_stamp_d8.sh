#!/usr/bin/env bash
# Lab build of libmgp_hip.so with -DMGP_D8_STAMP (spmm.hip): wave 0 of workgroup 0 of spmm_dict8_kernel records (phase, wall clock)
# pairs in an explicit buffer (tools/lab/stamp_d8.py).  Output: build_variants/stamp_d8/libmgp_hip.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
here="$root/manifold_gp_amd/csrc"
out="$root/build_variants/stamp_d8"; mkdir -p "$out"
bash "$here/build.sh" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I"$root/include" -I"$here" -DMGP_D8_STAMP \
    -c "$here/spmm.hip" -o "$out/spmm.o"
objs=$(ls "$here"/_obj/*.o | grep -v "/spmm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmgp_hip.so" "$out/spmm.o" $objs -L/opt/rocm/lib -lrccl
rm -f "$out"/*.o
echo "built $out/libmgp_hip.so"
