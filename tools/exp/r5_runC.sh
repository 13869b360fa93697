#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q -k "h16 or conv2d" > $O/pytest_h16.log 2>&1; echo "pytest h16 rc=$?"; tail -3 $O/pytest_h16.log
python -m pytest tests/test_parity_gpu.py -x -q -k "golden and full_t0" > $O/pytest_gold.log 2>&1; echo "pytest golden rc=$?"; tail -3 $O/pytest_gold.log
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --steps 50 --warmup 10"
for rep in 1 2 3; do
  python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean split-K epilogue', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee $O/bench3.txt
for rep in 1 2 3; do
  MADM_HIP_LIB=$R/build/libmadm_hip_before_sk.so python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('before               ', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
done | tee -a $O/bench3.txt
