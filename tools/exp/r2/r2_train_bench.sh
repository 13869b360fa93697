#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_trainbench; mkdir -p $O
cd $R
python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "train_step and not f32" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for dt in bf16 f16; do
timeout 900 python bench.py --workload train --steps 5 --warmup 2 --dtype $dt > $O/train_$dt.json 2> $O/train_$dt.err; echo "rc=$?"; cut -c1-1500 $O/train_$dt.json; tail -5 $O/train_$dt.err
done
