#!/bin/bash
# weight-gradient kernel: one vs two k-extents per staged step (MADM_WGRAD_SUB), per shape and in the training step
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad or backward" 2>&1 | tail -2
for sub in 1 2; do
  echo "== MADM_WGRAD_SUB=$sub"
  MADM_WGRAD_SUB=$sub timeout 300 python tools/bench_backward.py 2>&1 | grep -v amdgpu.ids | head -40
done
for sub in 1 2 ""; do
  echo "train step, MADM_WGRAD_SUB=${sub:-auto}: $(MADM_WGRAD_SUB=$sub python bench.py --workload train --steps 5 --warmup 2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms")')"
done
