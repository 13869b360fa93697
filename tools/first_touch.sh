#!/bin/bash
# Run FIRST in every gpurun call of the round: the box is fresh, so this is "the first GPU process on a fresh box" -- the one
# condition under which test_eval_forward_golden[f32-eval_depth] was once seen off by 3.7e-2 (profiles/round5_f32_eval_transient.txt).
# One line per box is appended to gpurun_out/first_touch/log.txt (merged back by gpurun); a failure keeps the full pytest output.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/first_touch; mkdir -p $O
cd $R
tag=$(date +%m%d_%H%M%S)_$(hostname | tail -c 7)
timeout 600 python -m pytest tests/test_eval_gpu.py -q -p no:cacheprovider -k "test_eval_forward_golden and f32 and eval_depth" > $O/$tag.out 2>&1
rc=$?
echo "$tag rc=$rc $(grep -E ' passed| failed' $O/$tag.out | tail -1)" >> $O/log.txt
[ $rc -eq 0 ] && rm -f $O/$tag.out
tail -1 $O/log.txt
