#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run21; mkdir -p $O
cd $R
export MADM_HIP_LIB=$R/build/libmadm_hip_stamps.so
(for i in 1 2; do
  python tools/exp/stamps_reg.py 64 320 2560 1 2 0
  python tools/exp/stamps_reg.py 64 320 2560 1 2 1
  NOBIAS=1 python tools/exp/stamps_reg.py 64 1280 320 1 2 0
done) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_reg2.txt
