#!/usr/bin/env bash
# PMC passes over the C == 1 SpMV only (tools/spmv_only.py); one counter group per pass.
set -o pipefail
failed=""      # passes that failed: their CSVs are removed (never summarised) and the script exits non-zero
out="gpurun_out/pmc_spmv"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" \
           "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 tools/spmv_only.py > "$out/p$i.log" 2>&1 || { echo "pass $i FAILED (rc $?)"; tail -5 "$out/p$i.log"; failed="$failed $i"; rm -rf "$out/p$i"; }
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_spmv/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spmv_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[5:] if len(v) > 10 else v
            print(d, k, "launches", len(v), "mean", sum(v) / max(1, len(v)))
PY
if [ -n "$failed" ]; then echo "failed passes:$failed"; exit 1; fi
