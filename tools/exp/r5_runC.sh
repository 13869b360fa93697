#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5C; mkdir -p $O
cd $R
python tools/tune_concurrent.py --workload train --streams 1 --max-m 100000000 --min-us 200 --rows $O/tuned_train.txt > $O/tune_train.txt 2>&1; grep -E "^1 |sums over" $O/tune_train.txt | head -40
for rep in 1 2; do
python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train table   ', d['ms_per_step'])"
MADM_TUNED_FILE=$O/tuned_train.txt python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train new rows', d['ms_per_step'])"
done | tee $O/ab_rows_train.txt
