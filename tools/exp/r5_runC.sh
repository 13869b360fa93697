#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5B; mkdir -p $O
cd $R
python -m pytest tests/test_eval_gpu.py -x -q > $O/pytest_eval.log 2>&1; echo "pytest eval rc=$?"; tail -2 $O/pytest_eval.log
python bench.py --workload eval --steps 20 --warmup 4 --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eval, new table', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
python tools/tune_concurrent.py --workload slide --max-m 100000000 --min-us 30 --rows $O/tuned_side_slide.txt > $O/tune_concurrent_slide.txt 2>&1; grep -E "^1 |sums over" $O/tune_concurrent_slide.txt | head -50
for rep in 1 2; do
python bench.py --workload slide --steps 10 --warmup 3 --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slide table   ', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
MADM_TUNED_FILE=$O/tuned_side_slide.txt python bench.py --workload slide --steps 10 --warmup 3 --no-kernel-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slide new rows', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/ab_rows_slide.txt
