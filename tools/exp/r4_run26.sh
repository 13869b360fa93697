#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run26; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "dwconv" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python tools/exp/bench_dwconv_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/dwconv_wgrad.txt
python bench.py --workload eval --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-profile 2>/dev/null | cut -c1-400 | tee $O/bench_eval.txt
python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | cut -c1-300 | tee $O/bench_train.txt
