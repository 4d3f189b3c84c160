#!/usr/bin/env bash
# Build libmgp_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${1:-$here/..}"
inc="$here/../../include"
obj="$here/_obj"
mkdir -p "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$inc -I$here -Wall -Wno-unused-function"
pids=()
for src in "$here"/*.hip; do
  o="$obj/$(basename "${src%.hip}").o"
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$inc/mgp_hip.h" -nt "$o" ] || [ "$here/mgp_common.h" -nt "$o" ] || [ "$here/mgp_internal.h" -nt "$o" ]; then
    extra=""
    # spmm.hip: keep the MFMA accumulators of spmm_mt_kernel in VGPRs (the default AGPR form copied 48 registers per loop
    # iteration between the two files of the unified register file); no other kernel of that file uses the matrix cores
    [ "$(basename "$src")" = spmm.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    $HIPCC $FLAGS $extra -c "$src" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libmgp_hip.so" "$obj"/*.o -L/opt/rocm/lib -lrccl
echo "built $out/libmgp_hip.so"
