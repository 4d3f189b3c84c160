#!/usr/bin/env bash
# Run on the GPU box (through gpurun): kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit
# one pass) of a bench command; tools/summarize_profile.py turns the output into profiles/<tag>_*.
# Usage: tools/profile.sh <tag> [bench.py arguments...]      e.g.  tools/profile.sh r02
#                                                                  tools/profile.sh r02_s5 --workload s5
# A failed pass (timeout-kill, non-zero profiler exit = a GPU hang or fault) makes the script exit non-zero.
set -o pipefail
tag="${1:-r02}"
shift || true
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
failed=0
run() {   # name, steps, warmup, profiler flags...
  local name="$1" steps="$2" warm="$3"; shift 3
  if ! timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps "$steps" --warmup "$warm" \
        --no-cpu-baseline --no-extras "${BENCH_ARGS[@]}" > "$out/bench_${name}.json" 2> "$out/bench_${name}.err"; then
    echo "pass $name FAILED (rc $?)"; tail -5 "$out/bench_${name}.err"; failed=1
  fi
}
BENCH_ARGS=("$@")
run trace 50 5 --kernel-trace --stats
[ "$failed" = 0 ] && run pmc_fetch 20 2 --pmc FETCH_SIZE
[ "$failed" = 0 ] && run pmc_write 20 2 --pmc WRITE_SIZE
find "$out" -name "*.csv" | head -20
exit $failed
