#!/bin/bash
# launch-strategy checks of bench.py
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "images/s", d["ms_per_step"], "ms; device", d["device_ms_per_step"], "serial", d["serial_ms_per_step"])'; }
echo "extract default: $(run)"
echo "extract pipeline 0 streams 3: $(run --pipeline 0 --streams 3)"
for w in eval slide; do for st in 1 2 3; do
echo "$w streams $st: $(run --workload $w --streams $st)"
done; done
