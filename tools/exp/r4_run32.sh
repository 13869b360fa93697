#!/bin/bash
# the VAE stem on the matrix pipe: correctness, kernel time, parity suite, bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run32; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "stem" > $O/pytest_stem.log 2>&1; echo "stem rc=$?"; tail -3 $O/pytest_stem.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/stem_time.txt
import os, torch
from madm_amd import ops
img = torch.rand((2, 3, 512, 512), device="cuda"); wT = torch.randn((27, 128), device="cuda") / 5; bias = torch.randn(128, device="cuda")
st = torch.zeros((2, 128, 2), dtype=torch.float64, device="cuda")
for kern in ("1", "0"):
    os.environ["MADM_STEM_KERNEL"] = kern
    for dt in (torch.float16, torch.bfloat16):
        f = lambda: ops.stem_conv3x3(img, wT, bias, dt, 0.5, 0.5, stats=st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print("kernel", "fma " if kern == "1" else "mfma", dt, f"{e0.elapsed_time(e1) / 10 * 1e3:.1f} us (incl. the range probe launch)")
PY
python -m pytest tests/test_parity_gpu.py -q -m gpu -x > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/pytest_parity.log
for i in 1 2; do
  MADM_STEM_KERNEL=1 python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_fma_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_mfma_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4run32/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['calib']['h16_128x128_512sq_us'])
PY
