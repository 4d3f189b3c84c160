#!/usr/bin/env bash
# Lab: a copy of the library with ONE source compiled under extra flags.  Usage: build_variant.sh <name> <file.hip> <flags...>
# Output: tools/lab/_kb_<name>/libmgp_hip.so (the other objects are the tree's: run manifold_gp_amd/csrc/build.sh first).
# Load it with MGP_LAB_LIB=... in the lab scripts that honour it (time_mt.py, kblock_shapes.py, ...) or _lib.LIB_PATH.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$here/../.."
name="$1"; file="$2"; shift 2
out="$here/_kb_$name"; mkdir -p "$out/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$root/include -I$root/manifold_gp_amd/csrc -Wno-unused-function"
extra=""
[ "$file" = spmm.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
base="$(basename "${file%.hip}")"
$HIPCC $FLAGS $extra "$@" -c "$root/manifold_gp_amd/csrc/$file" -o "$out/obj/$base.o"
objs=()
for o in "$root"/manifold_gp_amd/csrc/_obj/*.o; do
  [ "$(basename "$o")" = "$base.o" ] || objs+=("$o")
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libmgp_hip.so" "$out/obj/$base.o" "${objs[@]}" -L/opt/rocm/lib -lrccl
echo "built $out/libmgp_hip.so"
