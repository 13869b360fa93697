#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad or backward" 2>&1 | tail -2
timeout 300 python tools/bench_backward.py 2>&1 | grep -v amdgpu.ids | head -40
echo "train step: $(python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms")')"
