python tools/tune_insitu.py > gpurun_out/tune_insitu_extract4.txt 2>&1; sed -n 2,3p gpurun_out/tune_insitu_extract4.txt
python tools/tune_insitu.py --workload eval > gpurun_out/tune_insitu_eval4.txt 2>&1; sed -n 2,3p gpurun_out/tune_insitu_eval4.txt
