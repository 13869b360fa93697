#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "images/s", d["ms_per_step"], "ms; device", d["device_ms_per_step"], "serial", d["serial_ms_per_step"])'; }
for i in 1 2 3; do
echo "bracket on : $(run --pipeline 3 --hw-queues 8)"
echo "bracket off: $(MADM_BENCH_BRACKET=0 run --pipeline 3 --hw-queues 8)"
done
echo "bracket off, streams 3 whole-forward: $(MADM_BENCH_BRACKET=0 run --pipeline 0 --streams 3 --hw-queues 8)"
echo "bracket off, pipeline 3 q 16: $(MADM_BENCH_BRACKET=0 run --pipeline 3 --hw-queues 16)"
