#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O
cd $R
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2; do
for v in 256 128 0; do
  MADM_FUSE_GN_MAX_N=$v python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FUSE_GN_MAX_N=$v', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done
done | tee $O/ab_fuse_gn.txt
