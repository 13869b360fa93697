#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
python tools/exp/stress_eval_f32.py 400 f32 2>&1 | grep -v amdgpu.ids | tee $O/stress_f32_400.txt
python -m pytest tests/test_eval_gpu.py -q -x 2>&1 | tail -2 | tee $O/eval_suite_1.txt
python -m pytest tests/test_eval_gpu.py -q -x 2>&1 | tail -2 | tee $O/eval_suite_2.txt
