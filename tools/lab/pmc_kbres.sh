#!/usr/bin/env bash
# Lab: counters of kernel_block_res at 600 x 60000 x 100 with its stores (knob 5) and with every store dropped (knob 6):
# matrix-pipe busy cycles, wave wait cycles, vector-memory issue, L1 stalls.  One counter group per pass.
set -o pipefail
out="gpurun_out/pmc_kbres"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for knob in ${KNOBS:-5 6}; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
             "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_VMEM"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d "$out/k${knob}_p$i" -- python3 tools/lab/kblock_one.py 600 60000 100 $knob > "$out/k${knob}_p$i.log" 2>&1 || { echo "knob $knob pass $i FAILED"; tail -3 "$out/k${knob}_p$i.log"; rm -rf "$out/k${knob}_p$i"; }
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_kbres/k*_p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "kernel_block" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            v = v[10:]
            print(d.split("/")[-2], k, "launches", len(v), "mean", round(sum(v) / max(1, len(v)), 1))
PY
