#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "512 512 128 128" "256 256 256 256" "128 128 512 512"; do
  set -- $cfg
  for gn in "" "--gn"; do
    for extra in "" "--residual"; do
      timeout 120 python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 --tile 12 $gn $extra --rotate 4 --reps 30 --check 9 --dtype ${DT:-f16} 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
    done
  done
done
