#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5w; mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_poison_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -m pytest tests/test_train_gpu.py -x -q -k "matches_fixture and depth-f32" > $O/pytest_train.log 2>&1; echo "pytest train rc=$?"; tail -2 $O/pytest_train.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench_driver_flags.json
python bench.py --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=50', d['value'], d['ms_per_step'], d['serial_ms_per_step'])"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype --no-calib --pipeline 0 --streams 1 > $O/stats.log 2>&1)
python tools/last_replay.py $O/stats > $O/last_replay.txt; head -4 $O/last_replay.txt; grep -n "range_probe\|distribution" $O/last_replay.txt
