#!/bin/bash
# the profiling half of tools/round_validate.sh on the final tree (kernel statistics under both table profiles, the five PMC passes)
R=$GRAFT_REPO_ROOT; tag=r6v3; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
bash tools/first_touch.sh
python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile --pipeline 0 --streams 2 2>/dev/null | cut -c1-260
(cd /tmp && export TMPDIR=/tmp MADM_SYNC_PROFILE=throughput && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --pipeline 0 --streams 1 > $O/stats.log 2>&1)
python tools/kstats.py $O/stats 25 > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt; (python tools/last_replay.py $O/stats --expect 368 || true) > $O/last_replay.txt; head -8 $O/last_replay.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lat -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --pipeline 0 --streams 1 > $O/stats_lat.log 2>&1)
python tools/kstats.py $O/stats_lat 25 > $O/kernel_stats_latency.txt 2>&1; (python tools/last_replay.py $O/stats_lat --expect 377 || true) > $O/last_replay_latency.txt; head -8 $O/last_replay_latency.txt
export MADM_SYNC_PROFILE=throughput
bash tools/pmc.sh $tag/pmc $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --no-graph
unset MADM_SYNC_PROFILE
python tools/pmc_report.py $O/pmc --json $O/pmc_bench_extract.json | head -16
python tools/pmc_report.py $O/pmc "conv3x3_h|igemm_kernel|attn" > $O/pmc_report.txt 2>&1
rm -rf $O/stats/*/*trace* $O/stats_lat/*/*trace* 2>/dev/null
