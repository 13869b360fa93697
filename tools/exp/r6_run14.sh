#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6n; mkdir -p $O
cd $R
bash tools/first_touch.sh
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_driver_flags.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'serial', d['serial_ms_per_step'], 'alt', {k: v['value'] for k, v in d['alt_dtype'].items()}, 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('pmc', {}).get('mfma_util'), 'cpu', d['cpu_baseline']['value'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "suite rc=$? $(grep -E ' passed| failed' $O/pytest.log | tail -1)"
