#!/usr/bin/env bash
# Run on the GPU box (through gpurun): kernel-trace stats + two PMC passes of the default bench
# command, summaries copied to profiles/ by the caller.  Usage: tools/profile.sh <tag>
set -eo pipefail
tag="${1:-r01}"
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/bench_trace.json" 2> "$out/bench_trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > "$out/bench_fetch.json" 2> "$out/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > "$out/bench_write.json" 2> "$out/bench_write.err"
find "$out" -name "*.csv" | head -20
