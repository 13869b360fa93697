#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5M; mkdir -p $O
cd $R
python -m pytest tests/test_parity_gpu.py -x -q -k "staged" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib"
for rep in 1 2; do
python bench.py $B --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=50 enc prio', d['value'], d['ms_per_step'])"
MADM_EXP_PRIO=0 python bench.py $B --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=50 equal   ', d['value'], d['ms_per_step'])"
python bench.py $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=20 enc prio', d['value'], d['ms_per_step'])"
MADM_EXP_PRIO=0 python bench.py $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=20 equal   ', d['value'], d['ms_per_step'])"
done | tee $O/ab_prio.txt
