#!/bin/bash
# HBM-traffic PMC passes only (FETCH_SIZE, WRITE_SIZE) for one command.  usage: tools/pmc_hbm.sh <outdir> <script args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$out/p3 -- python3 "$@" > $R/gpurun_out/$out/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$out/p4 -- python3 "$@" > $R/gpurun_out/$out/p4.log 2>&1
