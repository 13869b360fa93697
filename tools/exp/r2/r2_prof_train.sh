#!/bin/bash
# rocprofv3 kernel stats of the training step (GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_proftrain; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload train --steps 2 --warmup 1 > $O/stats.log 2>&1
cd $R
python tools/kstats.py $O/stats 45 > $O/kernel_stats.txt 2>&1; head -60 $O/kernel_stats.txt
rm -rf $O/stats
