#!/bin/bash
tag=${1:-r4g11}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -x -q -s -k "lora-f32" 2>&1 | grep -v "^$" | tail -8 | cut -c1-300
MADM_NO_FUSE_PROJ_OUT=1 python -m pytest tests/test_train_gpu.py -x -q -s -k "lora-f32" 2>&1 | grep -v "^$" | tail -5 | cut -c1-300
python -m pytest tests/test_train_gpu.py -x -q -k "fixture" 2>&1 | tail -4 | cut -c1-300
