#!/usr/bin/env bash
# Lab: what the filtered key pass (dist_mfma_kernel<true>) spends outside its MFMA loop: the same search with
# (1) no range atomic, (2) no log stores, (3) neither, (4) no epilogue at all.  CAUTION (docs/kernels/knn.md): with nothing stored the
# compiler drops the accumulators' MFMAs too -- variants 3 and 4 time the operand copies alone, not the MFMA loop; use the in-kernel
# stamps (-DMGP_KNN_LAB=8, MGP_KNN_STAMPS=1 tools/lab/knn_filter_time.py) for the split between loop and epilogue.  Results are wrong in every variant (rows fail
# over): only the key kernel's duration in the kernel trace is read.  Build: knn_filter_bounds.sh build (CPU box); run on the GPU box.
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/../.."
if [ "${1:-run}" = build ]; then
  for v in 1 2 3 4; do bash tools/lab/build_variant.sh knnlab$v knn_mfma.hip -DMGP_KNN_LAB=$v > /dev/null; done
  ls tools/lab/_kb_knnlab*/libmgp_hip.so
else
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  out=gpurun_out/kf; mkdir -p $out
  for v in 0 1 2 3 4; do
    lib=""; [ $v != 0 ] && lib="tools/lab/_kb_knnlab$v/libmgp_hip.so"
    MGP_LAB_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$v -- python3 tools/lab/knn_filter_time.py 1 > $out/lab$v.log 2>&1
    echo "MGP_KNN_LAB=$v:"; python3 tools/lab/kstats.py $out/p$v 3 | grep "dist_mfma_kernel<true>\|select"; rm -rf $out/p$v
  done
fi
