#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run31; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "wgrad or unet_backward or linear_backward or conv" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== previous build"; MADM_HIP_LIB=$R/build/libmadm_hip_full.so timeout 300 python tools/exp/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_prev.txt
echo "== three register stages"; timeout 300 python tools/exp/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_new.txt
