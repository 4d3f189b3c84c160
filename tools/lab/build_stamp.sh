#!/usr/bin/env bash
# Lab build of libmgp_hip.so with -DMGP_STAMP (cg.hip, spmm.hip): block 0 of every CG kernel leaves wall_clock64 stamps
# behind the plan's state words (tools/lab/stamp_solve.py reads them).  Output: build_variants/stamp/libmgp_hip.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
here="$root/manifold_gp_amd/csrc"
out="$root/build_variants/stamp"; mkdir -p "$out"
bash "$here/build.sh" > /dev/null
for f in cg spmm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I"$root/include" -I"$here" -DMGP_STAMP \
    -c "$here/$f.hip" -o "$out/$f.o" &
done
wait
objs=$(ls "$here"/_obj/*.o | grep -v "/cg.o\|/spmm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmgp_hip.so" "$out/cg.o" "$out/spmm.o" $objs -L/opt/rocm/lib -lrccl
rm -f "$out"/*.o
echo "built $out/libmgp_hip.so"
