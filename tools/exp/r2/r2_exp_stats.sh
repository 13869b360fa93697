#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/prof_bench.sh r2_exp_stats_on
MADM_EXP_NO_STATS=1 bash tools/prof_bench.sh r2_exp_stats_off
rm -rf gpurun_out/r2_exp_stats_on/stats gpurun_out/r2_exp_stats_off/stats
