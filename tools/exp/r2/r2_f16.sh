#!/bin/bash
# f16 mode validation on the GPU box: parity tests, precision table, bench bf16 vs f16
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_f16; mkdir -p $O
cd $R
python -m pytest tests/test_parity_gpu.py tests/test_eval_gpu.py -m gpu -x -q -k "f16 or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -5 $O/pytest.log
python tools/precision_report.py > $O/precision.txt 2>&1; cat $O/precision.txt
python bench.py --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_bf16.json 2> $O/bench_bf16.err; cut -c1-400 $O/bench_bf16.json
python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dtype f16 > $O/bench_f16.json 2> $O/bench_f16.err; cut -c1-400 $O/bench_f16.json
