#!/usr/bin/env bash
# Lab (GPU box): device idle time inside a training epoch -- kernel trace of tools/profile_training.py, then tools/lab/epoch_gaps.py
# (kernel time, idle time by gap size, the kernel pairs that carry it).  Usage: tools/lab/epoch_gaps.sh [sup|semisup ...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
for mode in "${@:-semisup sup}"; do
  for m in $mode; do
    d=gpurun_out/prof_gap_$m
    timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/profile_training.py $m 5 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    grep '"mode"' $d.log | tail -1
    python3 tools/lab/epoch_gaps.py $d > gpurun_out/lab/gaps_$m.txt && cat gpurun_out/lab/gaps_$m.txt
    rm -rf $d
  done
done
