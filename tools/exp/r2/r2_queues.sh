#!/bin/bash
cd $GRAFT_REPO_ROOT
for q in 2 4 5 8; do for d in 0 1 2; do
  echo "GPU_MAX_HW_QUEUES=$q, $d idle streams first:"
  GPU_MAX_HW_QUEUES=$q timeout 200 python tools/exp/unet_concurrency.py --kmax 4 --rounds 40 --dummy $d --only 2,3,4 2>&1 | grep pipeline
done; done
