#!/usr/bin/env bash
# MFMA utilisation counters of the kernel-block kernel (tools/tune_kblock.py), one group per pass.
set -o pipefail
failed=""      # passes that failed: their CSVs are removed (never summarised) and the script exits non-zero
out="gpurun_out/pmc_mfma"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 tools/tune_kblock.py > "$out/p$i.log" 2>&1 || { echo "pass $i FAILED (rc $?)"; tail -5 "$out/p$i.log"; failed="$failed $i"; rm -rf "$out/p$i"; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_mfma/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "kernel_block" in r["Kernel_Name"]:
                acc[(r["Counter_Name"], r["Grid_Size"] if "Grid_Size" in r else "")].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k[0], "grid", k[1], "launches", len(v), "mean", sum(v) / len(v))
PY
if [ -n "$failed" ]; then echo "failed passes:$failed"; exit 1; fi
