"""Multi-GPU plumbing of the feature-extract path: one process per GPU, images sharded batch-wise,
NO data-path collective (replicas only -- GroupNorm/LayerNorm are per-sample, the
reference's evaluator does not reduce across ranks either, evaluation/d2_evaluator.py:228-238).
torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU) is used for the barrier and the
max-over-ranks of the timed region only."""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Initialises the default process group from the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of ``n_items`` images for ``rank`` (total_batch_size // world_size per rank
    as in data/build.py:77-90, remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX-reduce of a python float over the ranks (the bench's timed region)."""
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def sum_over_ranks(value, dist=None, device="cpu"):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()
