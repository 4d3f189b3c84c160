#!/usr/bin/env bash
# PMC passes over the kernels whose name contains <substr> in a python tool run (one counter group per pass).
# Usage (GPU box): tools/pmc_kernel.sh <tag> <kernel-substr> <script.py> [args...]
set -o pipefail
tag="$1"; sub="$2"; shift 2
failed=""
out="gpurun_out/pmc_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i FAILED (rc $?)"; tail -5 "$out/p$i.log"; failed="$failed $i"; rm -rf "$out/p$i"; }
  echo "pass $i done"
done
SUB="$sub" OUT="$out" python3 - <<'PY'
import csv, glob, collections, os
sub, out = os.environ["SUB"], os.environ["OUT"]
for d in sorted(glob.glob(out + "/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = v[5:] if len(v) > 10 else v
            print(k[0], k[1], "launches", len(v), "mean", round(sum(v) / max(1, len(v)), 1))
PY
if [ -n "$failed" ]; then echo "failed passes:$failed"; exit 1; fi
