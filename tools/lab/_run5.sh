set -o pipefail
python tools/lab/cg12.py 12 1 8 1 2>&1 | grep -v "amdgpu.ids\|\[bench\]" && python tools/lab/cg12.py 12 1 8 0 2>&1 | grep -v "amdgpu.ids\|\[bench\]" && python tools/lab/cg12.py 5 1 8 0 2>&1 | grep -v "amdgpu.ids\|\[bench\]" && python tools/lab/cg12.py 16 1 8 1 2>&1 | grep -v "amdgpu.ids\|\[bench\]" && python tools/lab/cg12.py 2 1 8 1 2>&1 | grep -v "amdgpu.ids\|\[bench\]" &&
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cg" 2>&1 | tail -5 &&
python tools/lab/semisup_breakdown.py semisup 5 2>&1 | grep -v amdgpu.ids | head -2
