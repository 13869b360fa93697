#!/bin/bash
# round 4, first evidence call: baseline bench on this box, ordered kernel listing of one replay, L2 hit-rate counters of
# the igemm classes, the A-stationary kernel alone vs in the replay, side-by-side tile tuning.
tag=${1:-r4e1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --pipeline 0 --streams 1 > $O/stats.log 2>&1)
python tools/last_replay.py $O/stats stem_conv3x3 . > $O/last_replay_ordered.txt; head -40 $O/last_replay_ordered.txt
rm -rf $O/stats
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $O/counters_all.txt 2>&1); grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" $O/counters_all.txt | sort -u | tr '\n' ' ' | cut -c1-3000
cd /tmp && export TMPDIR=/tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-graph --pipeline 0 --streams 1 > $O/pmc_$n.log 2>&1
  echo "pmc $n rc=$?"
done
cd $R
python tools/pmc_l2.py $O > $O/l2_hit_rates.txt 2>&1; head -60 $O/l2_hit_rates.txt
rm -rf $O/pmc_*/*/*kernel_trace* 2>/dev/null
python tools/bench_gemm.py --tile 13 --only 64 > $O/apanel_microbench.txt 2>&1; cat $O/apanel_microbench.txt
python tools/tune_concurrent.py --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; tail -50 $O/tune_concurrent.txt
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_tuned_side.json 2> $O/bench_tuned_side.err; cut -c1-300 $O/bench_tuned_side.json
python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/bench_again.json 2>/dev/null; cut -c1-300 $O/bench_again.json
