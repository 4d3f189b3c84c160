#!/usr/bin/env bash
# Lab: key-kernel variants of the filtered search (kernel-trace durations).  Usage (GPU box): knn_variants.sh <name>...  (tools/lab/_kb_<name>)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kf; mkdir -p $out
for v in production "$@"; do
  lib=""; [ $v != production ] && lib="tools/lab/_kb_$v/libmgp_hip.so"
  MGP_LAB_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pv -- python3 tools/lab/knn_filter_time.py 1 > $out/var_$v.log 2>&1
  echo "$v: $(grep 'search ms' $out/var_$v.log | tail -2 | cut -c1-16 | tr '\n' ' ')"; python3 tools/lab/kstats.py $out/pv 6 | grep "dist_mfma\|select\|bound\|regroup" | cut -c1-40,72-140; rm -rf $out/pv
done
