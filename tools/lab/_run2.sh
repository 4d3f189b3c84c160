set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
for mode in semisup sup; do
d=gpurun_out/prof_gap_$mode
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/profile_training.py $mode 5 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
grep '"mode"' $d.log | tail -1
python3 tools/lab/epoch_gaps.py $d > gpurun_out/lab/gaps_$mode.txt && cat gpurun_out/lab/gaps_$mode.txt
rm -rf $d
done
