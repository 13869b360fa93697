#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5z; mkdir -p $O
cd $R
python -m pytest tests/test_parity_gpu.py tests/test_eval_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2 3; do
python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new table', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/bench3.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver flags', d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['roofline']['frac'], d['calib']['h16_128x128_512sq_us'], d['alt_dtype'])" | tee -a $O/bench3.txt
