#!/usr/bin/env bash
# Lab: a copy of the library with features.hip compiled under extra -D flags.  Usage: build_kb_variant.sh <name> <-Dflags...>
# Output: tools/lab/_kb_<name>/libmgp_hip.so (the other objects are the tree's: run manifold_gp_amd/csrc/build.sh first)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$here/../.."
name="$1"; shift
out="$here/_kb_$name"; mkdir -p "$out/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$root/include -I$root/manifold_gp_amd/csrc -Wno-unused-function"
$HIPCC $FLAGS "$@" -c "$root/manifold_gp_amd/csrc/features.hip" -o "$out/obj/features.o"
objs=()
for o in "$root"/manifold_gp_amd/csrc/_obj/*.o; do
  [ "$(basename "$o")" = features.o ] || objs+=("$o")
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out/libmgp_hip.so" "$out/obj/features.o" "${objs[@]}" -L/opt/rocm/lib -lrccl
echo "built $out/libmgp_hip.so"
