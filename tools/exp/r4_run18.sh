#!/bin/bash
# head-GEMM L2 counters + the lora f32 fixture test after the near-tie bookkeeping change
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run18; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -q -m gpu -k "fixture" -x > $O/pytest_fixture.log 2>&1; echo "fixture rc=$?"; tail -3 $O/pytest_fixture.log
bash tools/exp/r4_headgemm_pmc.sh r4run18
