#!/bin/bash
# round 4: side-by-side tuning over every conv / GEMM class of the path, then the bench with those rows (A/B in one call).
tag=${1:-r4t2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python -m pytest tests/test_parity_gpu.py -x -q -k "staged_pipeline" > $O/pytest_pipe.log 2>&1; echo "pytest staged rc=$?"; tail -3 $O/pytest_pipe.log
python -m pytest tests/test_ops_gpu.py -x -q -k "post_groupnorm" > $O/pytest_pgn.log 2>&1; echo "pytest post-gn rc=$?"; tail -2 $O/pytest_pgn.log
python tools/tune_concurrent.py --max-m 100000000 --min-us 15 --rows $O/tuned_side.txt > $O/tune_concurrent.txt 2>&1; grep -c . $O/tuned_side.txt
B="--no-cpu-baseline --no-kernel-profile --no-alt-dtype"
python bench.py $B > $O/bench_base.json 2>/dev/null; cut -c1-200 $O/bench_base.json
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B > $O/bench_side.json 2>/dev/null; cut -c1-200 $O/bench_side.json
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B --pipeline 4 > $O/bench_side_p4.json 2>/dev/null; cut -c1-200 $O/bench_side_p4.json
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B --pipeline 2 > $O/bench_side_p2.json 2>/dev/null; cut -c1-200 $O/bench_side_p2.json
python bench.py $B > $O/bench_base2.json 2>/dev/null; cut -c1-200 $O/bench_base2.json
MADM_TUNED_FILE=$O/tuned_side.txt python bench.py $B --batch 4 --steps 25 > $O/bench_side_b4.json 2>/dev/null; cut -c1-200 $O/bench_side_b4.json
