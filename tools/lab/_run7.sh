set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cg or solve or train or schur or slq" 2>&1 | tail -3 || exit 1
for c in 12; do
bash tools/lab/trace_script.sh q1 tools/lab/cg12.py $c 1 8 1 > gpurun_out/q1_$c.txt 2>&1
grep "'C'" gpurun_out/q1_$c.txt; grep "calls" gpurun_out/q1_$c.txt | head -3
bash tools/lab/trace_script.sh q1 tools/lab/cg12.py $c 1 8 0 > gpurun_out/q1_$c.txt 2>&1
grep "'C'" gpurun_out/q1_$c.txt; grep "calls" gpurun_out/q1_$c.txt | head -3
MGP_UPD_QUADS=0 bash tools/lab/trace_script.sh q1 tools/lab/cg12.py $c 1 8 0 > gpurun_out/q1_$c.txt 2>&1
grep "'C'" gpurun_out/q1_$c.txt; grep "calls" gpurun_out/q1_$c.txt | head -3
done
python tools/lab/semisup_breakdown.py semisup 5 2>&1 | grep -v amdgpu.ids | head -2
