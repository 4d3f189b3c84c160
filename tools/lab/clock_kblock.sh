#!/usr/bin/env bash
# Effective clock and MFMA-pipe share of mgp_kernel_block's long launches (tools/lab/clock_kblock.py), random and zero operands.
set -o pipefail
out="gpurun_out/clock_kblock"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for z in 0 1; do
  export KB_ZERO=$z
  timeout -k 10 200 python3 tools/lab/clock_kblock.py > "$out/plain_$z.txt" 2>&1 || { echo "plain run failed"; tail -5 "$out/plain_$z.txt"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/p$z" -- python3 tools/lab/clock_kblock.py > "$out/p$z.log" 2>&1 || { echo "pmc run failed"; tail -5 "$out/p$z.log"; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for z in (0, 1):
    print(open("gpurun_out/clock_kblock/plain_%d.txt" % z).read().strip().replace("\n", "; "))
    dur = {}
    for f in glob.glob("gpurun_out/clock_kblock/p%d/**/*kernel_trace.csv" % z, recursive=True):
        for r in csv.DictReader(open(f)):
            if "kernel_block" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:])
    cnt = collections.defaultdict(dict)
    for f in glob.glob("gpurun_out/clock_kblock/p%d/**/*counter_collection.csv" % z, recursive=True):
        for r in csv.DictReader(open(f)):
            if "kernel_block" in r["Kernel_Name"]:
                cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    rows = collections.defaultdict(list)
    for d, (ns, name) in dur.items():
        c = cnt.get(d, {})
        if "GRBM_GUI_ACTIVE" in c and ns > 1e6:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            rows[name].append((ns / 1e3, cyc / ns, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc))
    for name, v in rows.items():
        v = v[2:] if len(v) > 4 else v
        n = len(v)
        print("  %s operands, %-34s: %d launches, %.0f us, clock %.2f GHz, MFMA pipe busy %.1f %% of those cycles" % (
            "zero" if z else "random", name, n, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, 100 * sum(x[2] for x in v) / n))
PY
rm -rf "$out"/p0 "$out"/p1
