#!/bin/bash
# eight-slot ring (tile 18) for the weight-streaming small-M layers: unit cases, side-by-side tuning of the M <= 2048 classes, bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5x; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/tune_concurrent.py --max-m 2048 --min-us 8 --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; grep -E "t18|sums over|^1 " $O/tune_concurrent.txt | head -60
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2; do
python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('table   ', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new rows', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/ab_rows.txt
