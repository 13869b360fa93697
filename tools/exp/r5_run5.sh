#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
python tools/exp/stress_eval_f32.py 40 f32 2>&1 | grep -v amdgpu.ids | tee $O/stress_f32.txt
MADM_HIP_LIB=$R/build/libmadm_hip_nopk.so python tools/exp/stress_eval_f32.py 40 f32 2>&1 | grep -v amdgpu.ids | tee $O/stress_f32_nopk.txt
python tools/exp/stress_eval_f32.py 40 f16 2>&1 | grep -v amdgpu.ids | tee $O/stress_f16.txt
for i in 1 2 3 4 5 6; do python -m pytest tests/test_eval_gpu.py -q -x -k "eval_forward_golden and f32" 2>&1 | tail -1; done | tee $O/eval_f32_repeat.txt
