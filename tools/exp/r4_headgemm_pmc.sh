#!/bin/bash
# L2 hit rate of the head's M = 524 288 GEMM (eval workload): tools/exp/head_gemm.py under rocprofv3 --pmc
tag=${1:-r4hg}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/pmc_TCC_HIT_sum -- python3 $R/tools/exp/head_gemm.py > $O/head_gemm.log 2>&1
echo "rc=$?"; tail -12 $O/head_gemm.log
cd $R; python tools/pmc_l2.py $O > $O/l2_head_gemm.txt 2>&1; head -30 $O/l2_head_gemm.txt
rm -rf $O/pmc_*/*/*kernel_trace* 2>/dev/null
