set -o pipefail
python -m pytest tests/ -x -q -m gpu -k "grad or train or semisup or schur or likelihood or autograd" 2>&1 | tail -5 || exit 1
python tools/lab/determinism.py sup 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2; do
python tools/lab/semisup_breakdown.py sup 8 2>&1 | grep -v amdgpu.ids | head -1
python tools/lab/semisup_breakdown.py semisup 8 2>&1 | grep -v amdgpu.ids | head -1
done
