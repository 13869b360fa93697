#!/bin/bash
# Round 6, VERDICT item 2 probe: do the two resident workgroups of the 16 x 16 halo conv run in lockstep?  MADM_H16_STAGGER=<clocks>
# delays the workgroup in the CU's upper LDS allocation in the first round.  (a) the LDS_ALLOC register values the first 1024 blocks
# see (stamps build), (b) layers alone per stagger, (c) the bench at the best values.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s; mkdir -p $O
cd $R
[ -z "$SKIP_FIRST" ] && bash tools/first_touch.sh
MADM_HIP_LIB=$R/build/libmadm_hip_h16stamps.so DT=f16 python tools/exp/stamps_h16.py 128 128 512 1 0 --lds-alloc 2>&1 | grep -v amdgpu.ids | tee $O/lds_alloc.txt
for shape in "512 512 128 128 --gn" "512 512 128 128" "512 512 128 128 --gn --residual" "256 256 256 256 --gn" "128 128 512 512 --gn" "256 256 128 256"; do
  set -- $shape
  for st in 0 1500 3000 5000 7000 9000 12000 16000; do
    echo -n "stagger $st: "
    MADM_H16_STAGGER=$st python tools/bench_one.py --hw $1 $2 --cin $3 --cout $4 ${@:5} --tile 12 --dtype f16 --graph --reps 20 --rotate 4 2>&1 | grep "TF/s"
  done
done | tee $O/layers.txt
for st in 0 5000 9000; do
  echo "== bench, stagger $st"
  MADM_H16_STAGGER=$st python bench.py --no-cpu-baseline --no-alt-dtype 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'serial', d.get('serial_ms_per_step'), 'roofline', d['roofline']['achieved'], d['roofline']['frac'])"
done | tee $O/bench.txt
