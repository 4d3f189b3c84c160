#!/usr/bin/env bash
set -o pipefail
out="gpurun_out/trace_multirhs"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p" -- python3 tools/lab/trace_multirhs.py > "$out/log.txt" 2>&1 || { tail -5 "$out/log.txt"; exit 1; }
tail -n 2 "$out/log.txt"
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/trace_multirhs/p/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print("%-90s calls %6s total %9.3f ms avg %9.2f us" % (r["Name"].replace("(anonymous namespace)::", "")[:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf "$out/p"
