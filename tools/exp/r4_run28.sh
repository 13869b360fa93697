#!/bin/bash
# kernel-argument touch in gn_apply / split-K reduce kernels: A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run28; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  MADM_HIP_LIB=$R/build/libmadm_hip_full.so python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_full_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_new_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4run28/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['calib']['h16_128x128_512sq_us'])
PY
