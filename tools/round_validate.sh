#!/bin/bash
# Driver-like validation on the GPU box: tests, smoke, bench lines of every workload, rocprofv3 kernel stats, PMC passes,
# precision table, DDP check.  usage (through gpurun): bash tools/round_validate.sh <tag>
tag=${1:-r3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
bash tools/first_touch.sh; python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt; cut -c1-900 $O/bench.json
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> /dev/null; cut -c1-200 $O/bench_bf16.json
python bench.py --workload eval --steps 10 --warmup 2 > $O/bench_eval.json 2> $O/bench_eval.err; cut -c1-300 $O/bench_eval.json
python bench.py --workload slide --steps 5 --warmup 2 > $O/bench_slide.json 2> $O/bench_slide.err; cut -c1-300 $O/bench_slide.json
python bench.py --workload train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err; cut -c1-300 $O/bench_train.json
python tools/precision_report.py > $O/precision.txt 2>&1; tail -3 $O/precision.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_check.py > $O/ddp_check_raw.txt 2>&1; grep -v "^\[W\|Gloo\|amdgpu.ids" $O/ddp_check_raw.txt > $O/ddp_check.txt; tail -2 $O/ddp_check.txt
python tools/bench_optim.py > $O/optim.txt 2>&1; cat $O/optim.txt
# kernel statistics of record: the one-stream graph under the THROUGHPUT rows of the tile table (MADM_SYNC_PROFILE=throughput: the kernels of the
# timed region and of bench.py's `roofline` / `kernels`), then the same under the latency profile (what a synchronous forward() runs: serial_*)
(cd /tmp && export TMPDIR=/tmp MADM_SYNC_PROFILE=throughput && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --pipeline 0 --streams 1 > $O/stats.log 2>&1)
python tools/kstats.py $O/stats 25 > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt; (python tools/last_replay.py $O/stats --expect 368 || true) > $O/last_replay.txt; head -30 $O/last_replay.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lat -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --pipeline 0 --streams 1 > $O/stats_lat.log 2>&1)
python tools/kstats.py $O/stats_lat 25 > $O/kernel_stats_latency.txt 2>&1; (python tools/last_replay.py $O/stats_lat --expect 377 || true) > $O/last_replay_latency.txt; head -3 $O/last_replay_latency.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 $R/bench.py --workload train --steps 2 --warmup 1 > $O/stats_train.log 2>&1)
python tools/kstats.py $O/stats_train 3 40 > $O/kernel_stats_train.txt 2>&1; head -14 $O/kernel_stats_train.txt
export MADM_SYNC_PROFILE=throughput   # (the PMC passes profile eager forwards: pinned to the rows of the timed region)
bash tools/pmc.sh $tag/pmc $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --no-graph
unset MADM_SYNC_PROFILE
python tools/pmc_report.py $O/pmc --json $O/pmc_bench_extract.json | head -30
python tools/pmc_report.py $O/pmc "conv3x3_h|igemm_kernel|attn" > $O/pmc_report.txt 2>&1
rm -rf $O/stats_train/*/*trace* 2>/dev/null
