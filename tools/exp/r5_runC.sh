#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5y; mkdir -p $O
cd $R
python tools/tune_concurrent.py --max-m 100000000 --min-us 8 --only "^M(8192|4096|131072|32768|524288) .* k1 " --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; grep -v amdgpu $O/tune_concurrent.txt | tail -40
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2; do
python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('table   ', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new rows', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/ab_rows.txt
