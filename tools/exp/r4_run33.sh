#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run33; mkdir -p $O
cd $R
export MADM_HIP_LIB=$R/build/libmadm_hip_c3dstamps.so
(python tools/exp/stamps_c3d.py 64 320 320 10 0
 python tools/exp/stamps_c3d.py 64 320 320 10 1
 python tools/exp/stamps_c3d.py 32 640 640 10 0) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_c3d.txt
