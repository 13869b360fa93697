python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/tune_insitu.py > gpurun_out/tune_insitu_extract3.txt 2>&1; sed -n 2,3p gpurun_out/tune_insitu_extract3.txt
python bench.py --no-cpu-baseline --no-kernel-profile 2>&1 | tail -1 | cut -c60-200
