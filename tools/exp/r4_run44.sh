#!/bin/bash
# tile / split-K rows for the training step's own shapes (data gradients, head, decoder): lone launches on one stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run44; mkdir -p $O
cd $R
timeout 1150 python tools/tune_concurrent.py --workload train --streams 1 --max-m 100000000 --min-us 250 --rows $O/tuned_train.txt > $O/tune_train.txt 2>&1; echo "tune rc=$?"
tail -40 $O/tune_train.txt | cut -c1-180
python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('table   ', d['ms_per_step'])"
MADM_TUNED_FILE=$O/tuned_train.txt python bench.py --workload train --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new rows', d['ms_per_step'])"
