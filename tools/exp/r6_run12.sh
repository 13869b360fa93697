#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6l; mkdir -p $O
cd $R
bash tools/first_touch.sh
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['unit'], d['ms_per_step'], 'ms/step; serial', d.get('serial_ms_per_step'))"; }
for i in 1 2; do
python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile 2>$O/bench.err | show "extract, synchronous calls under the latency profile   "
MADM_SYNC_PROFILE=throughput python bench.py --no-cpu-baseline --no-alt-dtype --no-kernel-profile 2>/dev/null | show "extract, MADM_SYNC_PROFILE=throughput                 "
done | tee $O/latency_profile.txt
for i in 1 2; do
python bench.py --workload eval --steps 20 --warmup 4 --no-kernel-profile 2>/dev/null | show "eval, latency profile                    "
MADM_SYNC_PROFILE=throughput python bench.py --workload eval --steps 20 --warmup 4 --no-kernel-profile 2>/dev/null | show "eval, MADM_SYNC_PROFILE=throughput       "
done | tee -a $O/latency_profile.txt
for i in 1 2 3; do
python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | show "train, latency profile                  "
MADM_SYNC_PROFILE=throughput python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | show "train, MADM_SYNC_PROFILE=throughput     "
done | tee -a $O/latency_profile.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "suite rc=$? $(grep -E ' passed| failed' $O/pytest.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.log | head -30
