#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5F; mkdir -p $O
cd $R
for st in 4 5 6 8 4 5 6; do
  python bench.py --workload eval --steps 20 --warmup 4 --streams $st --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eval streams $st', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/ab_eval_streams2.txt
for st in 3 4 5; do
  python bench.py --workload slide --steps 10 --warmup 3 --streams $st --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slide streams $st', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee -a $O/ab_eval_streams2.txt
for p in 3 4; do
  python bench.py --pipeline $p --slots $((2*p)) --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('extract unet streams $p', d['value'], d['ms_per_step'])"
done | tee -a $O/ab_eval_streams2.txt
