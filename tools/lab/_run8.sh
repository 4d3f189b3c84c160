set -o pipefail
MGP_BUILD_CHAIN=1 python tools/lab/semisup_breakdown.py semisup 8 2>&1 | grep -v amdgpu.ids | head -3
python tools/lab/semisup_breakdown.py semisup 8 2>&1 | grep -v amdgpu.ids | head -3
MGP_BUILD_CHAIN=1 python tools/lab/semisup_breakdown.py semisup 8 2>&1 | grep -v amdgpu.ids | head -1
python tools/lab/semisup_breakdown.py semisup 8 2>&1 | grep -v amdgpu.ids | head -1
