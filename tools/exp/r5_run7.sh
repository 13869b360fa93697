#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2; do
for sl in 0 4 5 6; do
  python bench.py $B --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots $sl', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done
done | tee $O/ab_slots.txt
for sl in 0 5; do
  python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 20 --warmup 5 --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=20 slots $sl', d['value'], d['ms_per_step'])"
done | tee -a $O/ab_slots.txt
for st in 4; do
  python bench.py $B --pipeline 4 --slots 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4 streams 6 slots', d['value'], d['ms_per_step'])"
  python bench.py $B --pipeline 2 --slots 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 streams 4 slots', d['value'], d['ms_per_step'])"
done | tee -a $O/ab_slots.txt
