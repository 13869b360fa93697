#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2; do
for sl in 0 5 7 8 9 11; do
  python bench.py $B --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots $sl', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done
done | tee $O/ab_slots2.txt
