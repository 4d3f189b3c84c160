#!/usr/bin/env bash
# Lab: rocprofv3 kernel-trace durations of mgp_kernel_block at 600 x 60000 x 100 per knob (isolated kernel durations, to set
# beside the back-to-back event timing of tools/lab/kblock_shapes.py)
set -o pipefail
out="gpurun_out/trace_kbres"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for knob in ${KNOBS:-1 5 6}; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/k$knob" -- python3 tools/lab/kblock_one.py 600 60000 100 $knob > "$out/k$knob.log" 2>&1 || { echo "knob $knob FAILED"; tail -3 "$out/k$knob.log"; }
  grep "us per launch" "$out/k$knob.log"
  python3 - "$out/k$knob" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "kernel_block" in r["Kernel_Name"]]
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[10:])
    starts = sorted(int(r["Start_Timestamp"]) for r in rows[10:])
    gaps = [b - a for a, b in zip(starts, starts[1:])]
    print("  kernel-trace: n %d  duration min %.1f median %.1f max %.1f us; start-to-start median %.1f us" % (len(d), d[0] / 1e3, d[len(d) // 2] / 1e3, d[-1] / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
PY
done
