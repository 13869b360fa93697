"""Multi-GPU plumbing of the feature-extract path: one process per GPU, images sharded batch-wise,
NO data-path collective (replicas only -- GroupNorm/LayerNorm are per-sample, the
reference's evaluator does not reduce across ranks either, evaluation/d2_evaluator.py:228-238).
torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU) is used for the barrier and the
max-over-ranks of the timed region only.  The training path's single exchange step -- the gradient all-reduce --
is GradBucketReducer below."""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Initialises the default process group from the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1 and not os.environ.get("MADM_FORCE_PROCESS_GROUP"):   # (the switch: the N > 1 code path on one GPU)
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise RuntimeError("madm_amd.dist.init: WORLD_SIZE > 1 needs MASTER_PORT (torchrun sets it)")
        import socket                      # single forced rank: any free port, so two jobs on one host never meet
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
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


class GradBucketReducer:
    """The ONE exchange step of the training path: all-reduce(mean) of the gradients, here over
    the flat fp32 gradient buffer of ``optim.FlatParams`` in a few large contiguous buckets instead of DDP's per-tensor
    buckets -- xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound, so fewer and
    larger messages win; 64 Mi floats (256 MB) per bucket keeps 13 buckets in flight for the full fine-tune (3.46 GB)
    and a single one for the LoRA mode (~32 MB).  ``reduce_range`` may be called as soon as a span of the buffer is
    final (the explicit backward fills it back to front), the collectives run asynchronously on the backend's stream
    and ``finish`` waits for them and applies the 1 / world scale.  Replaces torch DistributedDataParallel
    (engine/defaults.py: create_ddp_model) for the flat-buffer layout; no-op for world size 1."""

    def __init__(self, flat_grad, dist=None, bucket_numel=64 << 20):
        assert flat_grad.dim() == 1 and flat_grad.is_contiguous()
        self.g, self.dist, self.bucket = flat_grad, dist, int(bucket_numel)
        self.world = 1 if dist is None else dist.get_world_size()
        self.handles = []
        self.done_lo = flat_grad.numel()

    def reduce_range(self, lo, hi):
        """Starts the all-reduce of g[lo:hi] in bucket-sized pieces (async)."""
        if self.dist is None or self.world == 1:
            return
        for a in range(lo, hi, self.bucket):
            b = min(a + self.bucket, hi)
            self.handles.append(self.dist.all_reduce(self.g[a:b], op=self.dist.ReduceOp.SUM, async_op=True))

    def reduce_tail(self, lo):
        """Everything from ``lo`` to the last span already handed over is final: reduce it (back-to-front use)."""
        if lo < self.done_lo:
            self.reduce_range(lo, self.done_lo)
            self.done_lo = lo

    def finish(self):
        """Reduces whatever has not been handed over yet, waits, and turns the sums into means."""
        self.reduce_tail(0)
        for h in self.handles:
            h.wait()
        self.handles = []
        self.done_lo = self.g.numel()
        if self.world > 1:
            self.g.mul_(1.0 / self.world)
