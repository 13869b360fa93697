#!/bin/bash
# round 4: side-by-side tuning on the final kernels (packed FP32 off, new attention), then the driver-like validation
tag=${1:-r4final}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python tools/tune_concurrent.py --max-m 100000000 --min-us 15 --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; cat $O/tuned_side.txt
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype --steps 40 --warmup 8"
python bench.py $B 2>/dev/null | cut -c1-160
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B 2>/dev/null | cut -c1-160
python bench.py $B 2>/dev/null | cut -c1-160
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B 2>/dev/null | cut -c1-160
