#!/bin/bash
# side-by-side re-tune of the 3x3 conv shapes on the final kernels (epilogues without waits between stores), then A/B of the rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run39; mkdir -p $O
cd $R
timeout 1500 python tools/tune_concurrent.py --only "k3 " --min-us 25 --rows $O/tuned_side.txt > $O/tune.txt 2>&1; echo "tune rc=$?"; cat $O/tuned_side.txt
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype"
for i in 1 2; do
  python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('table  ', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
  MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new rows', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done
