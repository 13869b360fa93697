#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest.log 2>&1
echo "suite rc=$? $(grep -E ' passed| failed' $O/pytest.log | tail -1)"; grep -E "adapter tensors|^FAILED|^ERROR" $O/pytest.log | head -30
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'serial', d.get('serial_ms_per_step'), 'alt', {k: v['value'] for k, v in d.get('alt_dtype', {}).items()}, 'roofline', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'])"
