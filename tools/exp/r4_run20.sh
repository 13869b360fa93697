#!/bin/bash
# block-level stamps of the register-staged 128x64 kernel (tile 2) and the LDS-DMA tiles on the GEGLU / QKV shapes of the 64^2 level
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run20; mkdir -p $O
cd $R
export MADM_HIP_LIB=$R/build/libmadm_hip_stamps.so
for g in 1 0; do
  python tools/exp/stamps_reg.py 64 320 2560 1 2 $g
  python tools/exp/stamps_reg.py 64 320 960 1 2 $g
done 2>&1 | grep -v amdgpu.ids | tee $O/stamps_reg.txt
for t in 8 17 16; do python tools/exp/stamps_glds.py 64 320 2560 1 $t 1; done 2>&1 | grep -v amdgpu.ids | tee $O/stamps_glds.txt
