#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k; mkdir -p $O
cd $R
bash tools/first_touch.sh
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "suite rc=$? $(grep -E ' passed| failed' $O/pytest.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.log | head -30
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['unit'], d['ms_per_step'], 'ms/step; serial', d.get('serial_ms_per_step'), (d.get('roofline') or {}).get('pmc'))"; }
python bench.py --no-cpu-baseline 2>$O/bench.err | tee $O/bench.json | show extract
MADM_SYNC_PROFILE=throughput python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile 2>/dev/null | show "extract (sync profile = throughput)"
python bench.py --workload eval --steps 20 --warmup 4 --no-kernel-profile 2>/dev/null | show eval
MADM_SYNC_PROFILE=throughput python bench.py --workload eval --steps 20 --warmup 4 --no-kernel-profile 2>/dev/null | show "eval (sync profile = throughput)"
python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | show train
MADM_SYNC_PROFILE=throughput python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | show "train (sync profile = throughput)"
timeout 1500 python tools/tune_concurrent.py --workload train --min-us 200 --alone-rows $O/alone_train.txt > $O/tune_train.txt 2>&1; echo "tuner train rc=$?"; grep "sums over" $O/tune_train.txt; wc -l $O/alone_train.txt
