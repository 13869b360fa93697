#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run45; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  MADM_HIP_LIB=$R/build/libmadm_hip_full.so python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('table before', d['ms_per_step'])"
  python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with train rows', d['ms_per_step'])"
done
python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('extract', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
