#!/usr/bin/env python3
"""Data-parallel training check of MadmTrainer under torchrun (one process per rank):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/ddp_check.py [--backend gloo|nccl] [--size 64] [--steps 2]

Every rank builds the RGB->Depth training model at a small input size (f32 compute), perturbs its parameters by its rank
(so the start-up broadcast is exercised), takes its own data shard, runs ``--steps`` optimisation steps and then all ranks
compare: parameters, BatchNorm running statistics are NOT compared (per-GPU statistics, never SyncBN), AdamW moments and
the reduced gradient must be BIT-IDENTICAL on all ranks.  With fewer GPUs than ranks the ranks share cuda:0 and the
backend must be gloo (RCCL cannot put two ranks on one device)."""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default=None)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    from madm_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else 0
    backend = args.backend or ("nccl" if ndev >= world else "gloo")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = mdist.init(backend, device if backend == "nccl" else None)
    from test_train_gpu import build_product_train
    from golden_util import TRAIN_CASE
    from madm_amd.train import MadmTrainer
    model = build_product_train(torch.float32, size=args.size)
    with torch.no_grad():       # EVERY parameter, the frozen EMA teacher included: DDP's start-up broadcast covers them all
        for p in model.parameters():
            p.add_(0.01 * rank)
    trainer = MadmTrainer(model, lr=1e-4, weight_decay=0.05, grad_clip=1.0, dist=dist, amp=False)
    trainer.reducer.bucket = 8 << 20            # the small model's buffer is ~3.5 GB: several buckets per backward
    g = torch.Generator().manual_seed(100 + rank)
    data = []
    for _ in range(2):
        lab = torch.randint(0, TRAIN_CASE["K"], (1, args.size // 8, args.size // 8), generator=g)
        lab = lab.repeat_interleave(8, 1).repeat_interleave(8, 2).long()
        data.append({"source_rgb": 255.0 * torch.rand((3, args.size, args.size), generator=g), "source_label": lab,
                     "target_second_modality": 255.0 * torch.rand((3, args.size, args.size), generator=g)})
    random.seed(7 + rank)
    np.random.seed(7 + rank)
    for it in range(args.steps):
        losses, norm, stepped = trainer.run_step(data)
        print(f"rank {rank} step {it}: loss {sum(losses.values()):.5f} grad norm {norm:.5f} stepped {stepped} "
              f"all-reduce exposed {trainer.last_allreduce_exposed_ms} ms, started during backward "
              f"{trainer.last_overlap_frac}", flush=True)
    ok = True
    if dist is not None:
        mine = {id(p) for p in trainer.opt.flat.params}
        frozen = torch.cat([p.detach().flatten().float() for p in model.parameters() if id(p) not in mine])
        for name, t in (("parameters", trainer.opt.flat.flat), ("gradient", trainer.opt.flat.grad), ("exp_avg", trainer.opt.m),
                        ("exp_avg_sq", trainer.opt.v), ("frozen parameters (EMA teacher, VAE)", frozen)):
            ref = t.clone()
            dist.broadcast(ref, src=0)
            same = bool(torch.equal(ref, t))
            ok &= same
            if rank != 0 or not same:
                print(f"rank {rank}: {name} identical to rank 0: {same}", flush=True)
        dist.barrier()
    print(f"rank {rank}: DDP CHECK {'PASSED' if ok else 'FAILED'} (backend {backend}, world {world}, device {dev_index})", flush=True)
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
