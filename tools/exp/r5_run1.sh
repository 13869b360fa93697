#!/bin/bash
# round 5, first GPU call: the product runners (distinct batches per submit, deferred range assert), bench through submit(inputs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
python -m pytest tests/test_parity_gpu.py -x -q -k "staged or fused_proj" > $O/pytest_pipe.log 2>&1; echo "pytest pipe rc=$?"; tail -5 $O/pytest_pipe.log
python -m pytest tests/test_eval_gpu.py -x -q -k "inference_on_dataset" > $O/pytest_eval.log 2>&1; echo "pytest eval rc=$?"; tail -5 $O/pytest_eval.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
for v in "" "MADM_EXP_NO_SYNC_INPUTS=1" "MADM_EXP_NO_RANGE=1" "MADM_EXP_NO_SYNC_INPUTS=1 MADM_EXP_NO_RANGE=1"; do
  for rep in 1 2; do
    env $v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
  done
done | tee $O/ab_submit.txt
python bench.py --workload eval --steps 10 --warmup 2 > $O/bench_eval.json 2> $O/bench_eval.err; cut -c1-300 $O/bench_eval.json; tail -2 $O/bench_eval.err
python bench.py --workload slide --steps 5 --warmup 2 > $O/bench_slide.json 2> $O/bench_slide.err; cut -c1-300 $O/bench_slide.json; tail -2 $O/bench_slide.err
