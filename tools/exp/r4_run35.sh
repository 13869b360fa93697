#!/bin/bash
# halo-DMA conv, 64-channel tile: one barrier per kernel row (six-slot weight ring): stamps, correctness, A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run35; mkdir -p $O
cd $R
(export MADM_HIP_LIB=$R/build/libmadm_hip_c3dstamps.so
 python tools/exp/stamps_c3d.py 64 320 320 10 0 | head -14
 python tools/exp/stamps_c3d.py 32 640 640 10 1 | head -2) 2>&1 | grep -v amdgpu.ids | tee $O/stamps_c3d.txt
python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  MADM_HIP_LIB=$R/build/libmadm_hip_full.so python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_full_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_row_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4run35/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['calib']['h16_128x128_512sq_us'])
PY
