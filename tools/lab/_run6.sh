set -o pipefail
bash tools/lab/trace_script.sh u1024 tools/lab/cg12.py 12 1 8 1 > gpurun_out/u1024.txt 2>&1 ; MGP_UPD_BLOCK=256 bash tools/lab/trace_script.sh u256 tools/lab/cg12.py 12 1 8 1 > gpurun_out/u256.txt 2>&1
grep "calls" gpurun_out/u1024.txt | head -6; grep "calls" gpurun_out/u256.txt | head -6
