#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run29; mkdir -p $O
cd $R
timeout 600 python tools/exp/bench_norm_bw.py 2>&1 | grep -v amdgpu.ids | tee $O/norm_bw.txt
