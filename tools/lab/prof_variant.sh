#!/usr/bin/env bash
# Lab (GPU box): kernel-trace stats of the C3 solve loop for a given build of the library.
# Usage: tools/lab/prof_variant.sh <tag> [path of libmgp_hip.so]
set -o pipefail
tag="$1"; libp="${2:-}"
out="gpurun_out/prof_variant_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/lab/ab_lib.py $libp > "$out/run.log" 2>&1 || { echo FAILED; tail -5 "$out/run.log"; exit 1; }
f=$(ls "$out"/trace/*/*_kernel_stats.csv | head -1)
echo "== $tag"; grep "solve best" "$out/run.log"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("   %-60s calls %6s  mean %8.1f ns" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"])))
PY
rm -rf "$out/trace"
