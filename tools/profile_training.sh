#!/usr/bin/env bash
# Run on the GPU box: kernel time / launches per training epoch (supervised and semi-supervised, 60k) -> $MGP_PROFILE_OUT/<tag>_training_*.json
set -o pipefail
tag="${1:-r05}"
out="gpurun_out/profiles_out"; mkdir -p "$out"; export MGP_PROFILE_OUT="$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
fail=0
for mode in sup semisup; do
  for e in 3 7; do
    d="gpurun_out/prof_train_${mode}_$e"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/profile_training.py $mode $e > "$d.log" 2>&1 || { tail -5 "$d.log"; fail=1; }
    grep '"mode"' "$d.log" | tail -1
  done
  [ $fail = 0 ] && python3 tools/summarize_training_profile.py "$tag" $mode gpurun_out/prof_train_${mode}_3 3 gpurun_out/prof_train_${mode}_7 7 | head -60
  rm -rf gpurun_out/prof_train_${mode}_3 gpurun_out/prof_train_${mode}_7
done
exit $fail
