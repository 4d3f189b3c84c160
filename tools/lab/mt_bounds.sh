#!/usr/bin/env bash
# Lab: what bounds spmm_mt_kernel at C = 128 on the 60k graph -- the same launch with (1) the image requests, (2) the X requests,
# (3) both answered with zeros without memory traffic (out-of-range buffer loads), (4) every X request sent to one cached row.
# Build here (CPU box): bash tools/lab/mt_bounds.sh build ; run on the GPU box: bash tools/lab/mt_bounds.sh run
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/../.."
if [ "${1:-run}" = build ]; then
  for v in 1 2 3 4; do bash tools/lab/build_variant.sh mtlab$v spmm.hip -DMGP_MT_LAB=$v > /dev/null; done
  ls tools/lab/_kb_mtlab*/libmgp_hip.so
else
  echo "production:"; python3 tools/lab/time_mt.py c3 64 128 2>&1 | grep "^C "
  for v in 1 2 3 4; do
    echo "MGP_MT_LAB=$v:"; MGP_LAB_LIB=tools/lab/_kb_mtlab$v/libmgp_hip.so python3 tools/lab/time_mt.py c3 64 128 2>&1 | grep "^C "
  done
fi
