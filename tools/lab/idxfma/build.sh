#!/usr/bin/env bash
# Lab: generate + build the three variants of the VGPR-indexed accumulation micro-benchmark (run_idxfma.py <mode>).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
for m in 0 1 2; do
  MODE=$m python "$here/gen_idxfma.py"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -Wno-unused-value -o "$here/libidxfma$m.so" "$here/idxfma$m.hip"
done
echo built
