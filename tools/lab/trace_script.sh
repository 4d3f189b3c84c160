#!/usr/bin/env bash
# Kernel-time summary of a python tool under rocprofv3 --kernel-trace --stats: tools/lab/trace_script.sh <tag> <script.py> [args...]
set -o pipefail
tag="$1"; shift
out="gpurun_out/trace_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p" -- python3 "$@" > "$out/log.txt" 2>&1 || { tail -5 "$out/log.txt"; exit 1; }
grep -v "rocprofv3\|amdgpu.ids" "$out/log.txt" | tail -n 12
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/p/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("all kernels: %.3f ms in %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
    for r in rows[:26]:
        print("%-86s calls %6s total %9.3f ms avg %9.2f us" % (r["Name"].replace("(anonymous namespace)::", "")[:86], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf "$out/p"
