#!/usr/bin/env bash
# HBM traffic counters of the k-NN select kernel (tools/lab/knn_filter_time.py: 60k x 784 self-searches, candidate lists), FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: they do not fit one pass; FETCH_SIZE is doubled for wide coalesced reads on gfx950).
set -o pipefail
failed=""
out="gpurun_out/pmc_select"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/p$i" -- python3 tools/lab/knn_filter_time.py 1 > "$out/p$i.log" 2>&1 || { echo "pass $i FAILED (rc $?)"; tail -5 "$out/p$i.log"; failed="$failed $i"; rm -rf "$out/p$i"; }
done
python3 - <<'PY'
import csv, glob, collections, statistics
for d in sorted(glob.glob("gpurun_out/pmc_select/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "select_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("select_kernel", k, "launches", len(v), "median KB", statistics.median(v), "max KB", max(v))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "select_kernel" in r["Kernel_Name"]]
        if dur:
            print("select_kernel durations: launches", len(dur), "median us", statistics.median(dur) / 1e3)
PY
rm -rf "$out"/p*/
if [ -n "$failed" ]; then echo "failed passes:$failed"; exit 1; fi
