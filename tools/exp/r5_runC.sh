#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5D; mkdir -p $O
cd $R
python tools/tune_concurrent.py --lora --max-m 100000000 --min-us 8 --only " k1 " --rows $O/tuned_lora.txt > $O/tune_lora.txt 2>&1; grep -E "^1 |sums over" $O/tune_lora.txt | head -40
B="--lora --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2; do
python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lora table   ', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
MADM_TUNED_FILE=$O/tuned_lora.txt python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lora new rows', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/ab_rows_lora.txt
