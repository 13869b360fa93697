#!/bin/bash
# rocprofv3 kernel trace of the default bench (graph replay) + last-replay summary.  usage: prof_bench.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-calib --streams 1 "$@" > $O/stats.log 2>&1)
tail -1 $O/stats.log | cut -c1-200
python3 $R/tools/last_replay.py $O/stats > $O/last_replay.txt; head -40 $O/last_replay.txt
