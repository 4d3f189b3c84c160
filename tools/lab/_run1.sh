set -o pipefail
mkdir -p gpurun_out/lab
python tools/lab/semisup_breakdown.py semisup 5 > gpurun_out/lab/semi_chain8.txt 2>&1 && tail -25 gpurun_out/lab/semi_chain8.txt &&
MGP_CHAIN_MIN_C=48 python tools/lab/semisup_breakdown.py semisup 5 > gpurun_out/lab/semi_chain48.txt 2>&1 && head -1 gpurun_out/lab/semi_chain48.txt &&
python tools/lab/cg12.py 12 1 8 1 && python tools/lab/cg12.py 12 2 8 1 && python tools/lab/cg12.py 12 1 48 1 && python tools/lab/cg12.py 12 1 8 0 && python tools/lab/cg12.py 12 2 8 0
