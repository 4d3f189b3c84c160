#!/usr/bin/env bash
# Lab (GPU box): L2 hit / miss / request counters of the C = 1 tile SpMV launched back to back (does the matrix stay in the XCDs'
# L2s between launches?) at several graph sizes: tools/lab/pmc_c1.sh [nodes ...]   (0 = the C3 graph, 60 000 nodes)
set -o pipefail
out=gpurun_out/pmc_c1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for nodes in "${@:-0}"; do
  export MGP_NODES=$nodes
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/p$nodes -- python3 tools/lab/spmm_one.py 1 0 > $out/p$nodes.log 2>&1 || { echo "nodes $nodes FAILED"; tail -3 $out/p$nodes.log; continue; }
  echo "== nodes $nodes: $(grep 'us per launch' $out/p$nodes.log | tail -1) $(grep -o 'N=[0-9]* M=[0-9]* nnz_padded=[0-9]*' $out/p$nodes.log | tail -1)"
  python3 - $out/p$nodes <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmv_tile_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[5:] if len(v) > 10 else v
        print("  ", k, "launches", len(v), "mean", round(sum(v) / max(1, len(v)), 1))
PY
  rm -rf $out/p$nodes
done
