#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 1500 python tools/tune_concurrent.py --min-us 20 --alone-rows $O/alone_extract.txt > $O/tune_extract.txt 2>&1; echo "tuner extract rc=$?"; tail -3 $O/tune_extract.txt | cut -c1-300
timeout 1200 python tools/tune_concurrent.py --workload eval --min-us 30 --alone-rows $O/alone_eval.txt > $O/tune_eval.txt 2>&1; echo "tuner eval rc=$?"; tail -3 $O/tune_eval.txt | cut -c1-300
wc -l $O/alone_extract.txt $O/alone_eval.txt
SKIP_FIRST=1 bash tools/exp/r6_run9.sh
