#!/usr/bin/env bash
# MFMA utilisation counters of the k-NN key kernel (four 60k x 784 self-searches of tools/lab/knn_filter_time.py: the filtered key
# pass dist_mfma_kernel<true> and the keys to the sampled points dist_mfma_kernel<false>), one counter group per pass.
set -o pipefail
failed=""      # passes that failed: their CSVs are removed (never summarised) and the script exits non-zero
out="gpurun_out/pmc_knn"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 tools/lab/knn_filter_time.py 1 > "$out/p$i.log" 2>&1 || { echo "pass $i FAILED (rc $?)"; tail -5 "$out/p$i.log"; failed="$failed $i"; rm -rf "$out/p$i"; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_knn/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "dist_mfma_kernel" in r["Kernel_Name"]:
                inst = "filtered_pass<true>" if "dist_mfma_kernel<true>" in r["Kernel_Name"] else "sample_keys<false>"
                acc[(inst, r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k[0], k[1], "grid", k[2], "launches", len(v), "mean", sum(v) / len(v))
PY
if [ -n "$failed" ]; then echo "failed passes:$failed"; exit 1; fi
