#!/bin/bash
# the whole GPU suite three times in three processes on one box: how often does anything fail?  (profiles/round5_f32_eval_transient.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5E; mkdir -p $O
cd $R
for i in 1 2 3; do
  python -m pytest tests -m gpu -q -x > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -E 'passed|failed' $O/pytest_$i.log | tail -1)"
done | tee $O/suite_repeat.txt
