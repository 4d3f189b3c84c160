#!/usr/bin/env bash
# Lab: LDS bank-conflict share per kernel of a python tool run (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE summed over its launches).
# Usage (GPU box): tools/lab/pmc_lds.sh <tag> <script.py> [args...]
set -o pipefail
tag="$1"; shift
out="gpurun_out/pmc_lds_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/p" -- python3 "$@" > "$out/run.log" 2>&1 || { echo "FAILED"; tail -5 "$out/run.log"; }
OUT="$out" python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(os.environ["OUT"] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))
for k, v in rows[:22]:
    a = v.get("SQ_LDS_IDX_ACTIVE", 0)
    if a > 0:
        print("%-62s launches %5d  lds cycles %12.0f  conflict %5.1f %%  lds insts %11.0f" % (k, n[k], a, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / a, v.get("SQ_INSTS_LDS", 0)))
PY
rm -rf "$out/p"
