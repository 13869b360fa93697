#!/bin/bash
# epilogue without waits between stores: stamps, correctness, same-box A/B of the bench against the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run22; mkdir -p $O
cd $R
(export MADM_HIP_LIB=$R/build/libmadm_hip_stamps.so
 python tools/exp/stamps_reg.py 64 320 2560 1 2 0
 python tools/exp/stamps_reg.py 64 320 2560 1 2 1
 python tools/exp/stamps_reg.py 64 1280 320 1 2 0) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_reg.txt
python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  MADM_HIP_LIB=$R/build/libmadm_hip_prev.so python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_prev_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_new_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4run22/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['calib']['h16_128x128_512sq_us'])
PY
