#!/bin/bash
# the driver's multi-GPU launch form, on one GPU: torchrun + RCCL init + barrier + max over ranks
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4run40; mkdir -p $O
cd $R
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-alt-dtype > $O/torchrun_n1.json 2> $O/torchrun_n1.err; echo "torchrun rc=$?"; cut -c1-300 $O/torchrun_n1.json; grep -v "amdgpu.ids\|^\[W" $O/torchrun_n1.err | tail -5
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --workload train --steps 3 --warmup 1 > $O/torchrun_train_n1.json 2> $O/torchrun_train_n1.err; echo "torchrun train rc=$?"; cut -c1-420 $O/torchrun_train_n1.json
python bench.py --gpus 2 --steps 5 --warmup 2 > $O/gpus2.json 2> $O/gpus2.err; echo "gpus2 rc=$? (expected: refuses, one device visible)"; tail -2 $O/gpus2.err | cut -c1-200
