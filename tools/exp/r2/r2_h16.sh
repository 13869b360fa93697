#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "512 512 128 128" "256 256 256 256" "128 128 512 512" "256 256 128 256" "64 64 512 512"; do
  set -- $cfg
  for gn in "" "--gn"; do
    for t in 9 12; do
      timeout 120 python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 --tile $t $gn --rotate 4 --reps 30 --check 9 --dtype ${DT:-bf16} 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
    done
  done
done
timeout 120 python tools/bench_one.py --hw 48 80 --cin 128 --cout 192 --tile 12 --gn --check 9 --dtype f32 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/bench_one.py --hw 40 24 --cin 64 --cout 64 --tile 12 --check 9 --dtype f32 --residual 2>&1 | grep -v amdgpu.ids
