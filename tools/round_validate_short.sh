#!/bin/bash
# tests + smoke + the bench line of every workload (no profiler passes): the last check after a late kernel change
tag=${1:-r2s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
bash tools/first_touch.sh; python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt; cut -c1-300 $O/bench.json
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> /dev/null; cut -c1-200 $O/bench_bf16.json
python bench.py --workload eval > $O/bench_eval.json 2> $O/bench_eval.err; cut -c1-200 $O/bench_eval.json
python bench.py --workload slide > $O/bench_slide.json 2> $O/bench_slide.err; cut -c1-200 $O/bench_slide.json
python bench.py --workload train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err; cut -c1-200 $O/bench_train.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/bench.py --workload train --steps 2 --warmup 1 > $O/stats_train.log 2>&1)
python tools/kstats.py $O/stats_train 3 40 > $O/kernel_stats_train.txt 2>&1; head -6 $O/kernel_stats_train.txt
rm -rf $O/stats_train
