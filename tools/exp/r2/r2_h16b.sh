#!/bin/bash
cd $GRAFT_REPO_ROOT
for sk in 0 -1 20000 40000 60000; do
  if [ $sk -ge 0 ]; then export MADM_H16_SKEW=$sk; else unset MADM_H16_SKEW; fi
  echo "== skew $sk"
  for cfg in "512 512 128 128" "256 256 256 256" "128 128 512 512"; do
    set -- $cfg
    for gn in "" "--gn"; do
      timeout 120 python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 --tile 12 $gn --rotate 4 --reps 30 --dtype ${DT:-bf16} 2>&1 | grep -v amdgpu.ids
    done
  done
done
