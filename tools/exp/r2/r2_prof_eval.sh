#!/bin/bash
# rocprofv3 kernel stats of the eval forward (whole-forward graph, one stream)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_profeval; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload eval --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-profile --pipeline 0 --streams 1 > $O/stats.log 2>&1
cd $R
python tools/last_replay.py $O/stats > $O/last_replay.txt 2>&1; head -40 $O/last_replay.txt
rm -rf $O/stats
