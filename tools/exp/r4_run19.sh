#!/bin/bash
# two-slot small tiles (16 / 17) + tuner with the layers' own epilogues (LayerNorm fold, GEGLU, residual) on the linear layers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run19; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "glds64d or glds128x64d" -x > $O/pytest_tiles.log 2>&1; echo "tiles rc=$?"; tail -3 $O/pytest_tiles.log
python -m pytest tests/test_train_gpu.py -q -m gpu -k "fixture and lora" -x > $O/pytest_fixture.log 2>&1; echo "fixture rc=$?"; tail -3 $O/pytest_fixture.log
timeout 1500 python tools/tune_concurrent.py --only "k1 s1" --min-us 40 --rows $O/tuned_side.txt > $O/tune.txt 2>&1; echo "tune rc=$?"
cat $O/tune.txt | cut -c1-200
