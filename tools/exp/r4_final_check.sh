#!/bin/bash
# the driver's round-end sequence on the final tree: GPU tests, smoke, bench with the driver's flags
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4finalcheck; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_driver_like.json')); print(d['value'], d['ms_per_step'], d['serial_ms_per_step'], d['calib'], d['value_normalised'], d['roofline']['frac'], d['cpu_baseline']['value'])"
