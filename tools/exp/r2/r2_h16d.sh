#!/bin/bash
# residual variants of the 16x16 halo conv: stamps + timing
cd $GRAFT_REPO_ROOT
if [ -f madm_amd/libmadm_hip_H16STAMPS.so ]; then
  for a in "128 128 512 0 1" "128 128 512 1 1" "128 128 512 0 0"; do
    MADM_HIP_LIB=$PWD/madm_amd/libmadm_hip_H16STAMPS.so DT=${DT:-f16} timeout 120 python tools/exp/stamps_h16.py $a 2>&1 | grep -v amdgpu.ids
  done
fi
for cfg in "512 512 128 128" "256 256 256 256" "128 128 512 512"; do
  set -- $cfg
  for gn in "" "--gn"; do
    for extra in "" "--residual"; do
      timeout 120 python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 --tile 12 $gn $extra --rotate 4 --reps 30 --check 9 --dtype ${DT:-f16} 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
    done
  done
done
