#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run27; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py tests/test_eval_gpu.py -q -m gpu -x -k "dwconv or spatial" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python tools/exp/bench_dwconv.py 2>&1 | grep -v amdgpu.ids | tee $O/dwconv.txt
timeout 300 python tools/exp/bench_dwconv_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/dwconv_wgrad.txt
