for i in 1 2 3; do
MGP_NO_REBIND=1 MGP_NO_REPEAT=1 python tools/lab/semisup_breakdown.py sup 8 2>&1 | grep -v amdgpu.ids | head -1
done
