#!/usr/bin/env bash
# Run on the GPU box (through gpurun): every committed profile of a round taken again on the current source tree, summaries
# under gpurun_out/profiles_out/<tag>_* (then, here: python tools/assemble_profiles.py <tag>, and commit profiles/).  Usage: tools/retake_profiles.sh r04
# Needed after ANY change to manifold_gp_amd/csrc/*.hip|*.h or include/*.h: bench.py quotes a profile only while its
# source_hash equals the tree's (roofline.profile_age_ok).
set -o pipefail
tag="${1:-r04}"
out="gpurun_out/profiles_out"; mkdir -p "$out"
export MGP_PROFILE_OUT="$out"
fail=0
bash tools/profile.sh "$tag" || fail=1
[ $fail = 0 ] && python3 tools/summarize_profile.py "$tag" | tail -3
[ $fail = 0 ] && python3 tools/trace_solve.py "gpurun_out/prof_${tag}/trace" > "$out/${tag}_trace_solve.txt"
[ $fail = 0 ] && { bash tools/profile.sh "${tag}_s5" --workload s5 --classic-cg || fail=1; }
[ $fail = 0 ] && python3 tools/summarize_profile.py "${tag}_s5" | tail -3
# the eigensolve's kernel mix (three 60k eigensolves of 100 pairs at the shipped tolerance)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ $fail = 0 ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_eig -- python3 tools/time_eigen.py 100 1e-6 > gpurun_out/prof_eig.log 2>&1 || fail=1
  f=$(ls -t gpurun_out/prof_eig/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "$out/${tag}_eigensolve_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:24]:
        w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
  grep eigensolve gpurun_out/prof_eig.log | tail -3
fi
# matrix-pipe counters of the kernel block, cache counters of the wide SpMM (matrix-core tiles against the gather kernel)
[ $fail = 0 ] && { bash tools/pmc_mfma.sh > "$out/${tag}_pmc_mfma_raw.txt" 2>&1 || fail=1; }
[ $fail = 0 ] && { bash tools/pmc_kernel.sh mt spmm_mt tools/lab/spmm_one.py 128 2 0 0 0 1 > "$out/${tag}_pmc_spmm_mt_raw.txt" 2>&1 || fail=1; }
[ $fail = 0 ] && { MGP_NO_CHAIN=1 bash tools/pmc_kernel.sh mtgiven spmm_mt tools/lab/spmm_one.py 128 2 0 0 0 1 > "$out/${tag}_pmc_spmm_mt_given_raw.txt" 2>&1 || fail=1; }
[ $fail = 0 ] && { bash tools/pmc_kernel.sh gather128 'spmm_kernel<64' tools/lab/spmm_one.py 128 0 0 0 0 0 > "$out/${tag}_pmc_spmm_gather_raw.txt" 2>&1 || fail=1; }
# the kernel block at the C3 posterior shape: matrix-pipe counters and kernel-trace durations, default kernel against the lean one
[ $fail = 0 ] && { KNOBS="0 1" bash tools/lab/pmc_kbres.sh > "$out/${tag}_pmc_kbres_raw.txt" 2>&1 || fail=1; }
[ $fail = 0 ] && { KNOBS="0 1 6" bash tools/lab/trace_kbres.sh > "$out/${tag}_trace_kbres_raw.txt" 2>&1 || fail=1; }
# round 5: the k-NN search (kernel mix of three 60k x 784 searches + the key kernel's matrix-pipe counters + the select kernel's HBM
# counters), the S5 solve as the library runs it by default (COCG on the complex factor: 4-column tile SpMM + cx_update), the
# training epochs' kernel time and launch count
if [ $fail = 0 ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_knn -- python3 tools/time_knn.py > gpurun_out/prof_knn.log 2>&1 || fail=1
  f=$(ls -t gpurun_out/prof_knn/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "$out/${tag}_knn_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:16]:
        w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
  grep "search ms" gpurun_out/prof_knn.log | head -3
fi
[ $fail = 0 ] && { bash tools/pmc_knn.sh > "$out/${tag}_pmc_knn_mfma_raw.txt" 2>&1 || fail=1; }
[ $fail = 0 ] && { bash tools/pmc_select.sh > "$out/${tag}_pmc_knn_select_raw.txt" 2>&1 || fail=1; }
if [ $fail = 0 ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s5cx -- python3 bench.py --workload s5 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/prof_s5cx.json 2> gpurun_out/prof_s5cx.err || fail=1
  f=$(ls -t gpurun_out/prof_s5cx/*/*_kernel_stats.csv | head -1)
  python3 - "$f" "$out/${tag}_s5_complex_shift_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:16]:
        w.writerow([r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
  cp gpurun_out/prof_s5cx.json "$out/${tag}_s5_complex_shift_bench.json"
fi
[ $fail = 0 ] && { bash tools/profile_training.sh "$tag" > "$out/${tag}_training_log.txt" 2>&1 || fail=1; }
rm -rf gpurun_out/pmc_kbres gpurun_out/trace_kbres gpurun_out/prof_knn gpurun_out/pmc_knn gpurun_out/pmc_select gpurun_out/prof_s5cx
# gpurun copies at most 64 MiB of gpurun_out/ back: keep the summaries, drop the raw traces
rm -rf gpurun_out/prof_"${tag}" gpurun_out/prof_"${tag}"_s5 gpurun_out/prof_eig gpurun_out/pmc_mfma gpurun_out/pmc_mt gpurun_out/pmc_mtgiven gpurun_out/pmc_gather128
ls -la "$out"
tail -n 12 "$out/${tag}_pmc_mfma_raw.txt"
exit $fail
