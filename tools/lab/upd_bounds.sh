#!/usr/bin/env bash
# Lab: what bounds cg_update_kernel at 2 <= C <= 16.  Builds (here, no GPU needed) three variants of the library with parts of the
# kernel removed (-DMGP_LAB_UPD: 4 = never converges, alpha = beta = 0 so the vectors stay finite; 5 = + no partial reads;
# 6 = + no vector pass; 7 = neither), then on the GPU box: tools/lab/upd_bounds.sh run  (kernel-trace mean of each).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
if [ "${1:-build}" = build ]; then
  for v in 4 5 6 7; do bash "$here/build_variant.sh" upd$v cg.hip -DMGP_LAB_UPD=$v; done
  exit 0
fi
cd "$here/../.."
for v in 4 5 6 7; do
  MGP_LAB_LIB=tools/lab/_kb_upd$v/libmgp_hip.so bash tools/lab/trace_script.sh upd$v tools/lab/cg12.py ${2:-12} 1 48 ${3:-0} > gpurun_out/upd$v.txt 2>&1 || true
  echo "== MGP_LAB_UPD=$v"; grep "cg_update_kernel\|spmm_tile_q" gpurun_out/upd$v.txt | head -3
done
