#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run37; mkdir -p $O
cd $R
timeout 600 python tools/exp/sweep_wgrad_splitm.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 | tee $O/sweep.txt
python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "wgrad or unet_backward or linear_backward or conv" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | cut -c1-300 | tee $O/bench_train.txt
