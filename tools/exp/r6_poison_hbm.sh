#!/bin/bash
# VERDICT r5 "do this" 1: (a) a fresh-box FIRST-process run of the test that once failed, (b) the whole GPU suite with every
# uninitialised HBM allocation poisoned (MADM_DEBUG_POISON_HBM=1: NaN, =2: huge finite), (c) N fresh processes of the f32 eval test.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O
cd $R
N=${1:-30}
bash tools/first_touch.sh
for mode in 1 2; do
  export MADM_DEBUG_POISON_HBM=$mode
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_poison_gpu.py > $O/suite_poison$mode.log 2>&1
  echo "poison mode $mode: rc=$? $(grep -E ' passed| failed' $O/suite_poison$mode.log | tail -1)" | tee -a $O/summary.txt
  grep -E "^(FAILED|ERROR)" $O/suite_poison$mode.log | head -40 | tee -a $O/summary.txt
done
unset MADM_DEBUG_POISON_HBM
for i in $(seq 1 $N); do
  timeout 300 python -m pytest tests/test_eval_gpu.py -q -p no:cacheprovider -k "test_eval_forward_golden and f32 and eval_depth" > $O/fresh_$i.log 2>&1
  echo "fresh process $i rc=$? $(grep -E ' passed| failed' $O/fresh_$i.log | tail -1)"
done | tee $O/fresh_processes.txt
echo "fresh processes failing: $(grep -c -v 'rc=0' $O/fresh_processes.txt) of $N" | tee -a $O/summary.txt
