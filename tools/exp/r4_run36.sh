#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run36; mkdir -p $O
cd $R
timeout 600 python tools/exp/sweep_wgrad_splitm.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
