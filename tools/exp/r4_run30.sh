#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run30; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "norm or groupnorm or batch or bn or stats or unet_backward or head" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/exp/bench_norm_bw.py 2>&1 | grep -v amdgpu.ids | tee $O/norm_bw.txt
python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | cut -c1-300 | tee $O/bench_train.txt
