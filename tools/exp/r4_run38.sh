#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run38; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "wgrad or unet_backward or linear_backward or conv or fixture" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
echo "== previous build"; MADM_HIP_LIB=$R/build/libmadm_hip_full.so timeout 300 python tools/exp/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-80 | tee $O/wgrad_prev.txt
echo "== two register sets, branch-free offsets"; timeout 300 python tools/exp/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-80 | tee $O/wgrad_new.txt
python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | cut -c1-300 | tee $O/bench_train.txt
