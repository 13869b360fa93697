#!/bin/bash
# rocprofv3 kernel trace of the STAGED pipeline (default bench launch): who runs beside whom.  usage (gpurun): bash tools/exp/pipeline_trace.sh <tag>
tag=${1:-r3_ptrace}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/trace.log 2>&1)
tail -1 $O/trace.log | cut -c1-200
python3 $R/tools/exp/pipeline_trace.py $O/trace > $O/pipeline_overlap.txt; cat $O/pipeline_overlap.txt
find $O/trace -name "*agent_info*" -delete
