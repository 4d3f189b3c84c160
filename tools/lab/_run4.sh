mkdir -p gpurun_out/lab
python tools/lab/epoch_cprofile.py semisup 6 > gpurun_out/lab/cprof_semisup.txt 2>&1 && python tools/lab/epoch_cprofile.py sup 6 > gpurun_out/lab/cprof_sup.txt 2>&1
