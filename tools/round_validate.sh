#!/bin/bash
# Driver-like validation on the GPU box: tests, smoke, bench line, rocprofv3 kernel stats, PMC passes.
# usage (through gpurun): bash tools/round_validate.sh <tag>
tag=${1:-r1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt; cut -c1-900 $O/bench.json
python bench.py --workload eval --steps 10 --warmup 2 > $O/bench_eval.json 2> $O/bench_eval.err; cut -c1-300 $O/bench_eval.json
python tools/bench_optim.py > $O/optim.txt 2>&1; cat $O/optim.txt
python tools/bench_backward.py > $O/backward_blocks.txt 2>&1; tail -14 $O/backward_blocks.txt
python tools/bench_unet_backward.py > $O/unet_backward.txt 2>&1; tail -4 $O/unet_backward.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --streams 1 > $O/stats.log 2>&1)
python tools/kstats.py $O/stats 25 > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt; python tools/last_replay.py $O/stats > $O/last_replay.txt; head -30 $O/last_replay.txt
bash tools/pmc.sh $tag/pmc $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-graph
python tools/pmc_report.py $O/pmc --json $O/pmc_bench_extract.json | head -30
python tools/pmc_report.py $O/pmc "conv3x3_halo|igemm_kernel|attn" > $O/pmc_report.txt 2>&1
