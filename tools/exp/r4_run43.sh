#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run43; mkdir -p $O
cd $R
export MADM_HIP_LIB=$R/build/libmadm_hip_h16stamps.so DT=f16
(python tools/exp/stamps_h16.py 128 128 512 1 0
 python tools/exp/stamps_h16.py 128 128 512 0 0
 python tools/exp/stamps_h16.py 128 128 512 1 1
 python tools/exp/stamps_h16.py 512 512 128 1 0) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_h16.txt
