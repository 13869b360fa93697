"""world_size-2 gloo test of the multi-GPU plumbing (madm_amd/dist.py): shards cover the batch exactly
once, the barrier + MAX reduction of the timed region and the SUM of processed units behave as bench.py
assumes.  Runs on CPU."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from madm_amd import dist as mdist
    d = mdist.init(backend="gloo")
    assert d is not None
    lo, hi = mdist.shard_batch(5, rank, world)
    d.barrier()
    t = mdist.max_over_ranks(1.0 + rank, d)
    n = mdist.sum_over_ranks(hi - lo, d)
    q.put((rank, lo, hi, t, n))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]       # shards tile [0, 5)
    assert all(r[3] == 2.0 for r in res)                          # MAX over ranks
    assert all(r[4] == 5.0 for r in res)                          # every image processed once


def test_shard_batch_properties():
    from madm_amd.dist import shard_batch
    for n in (0, 1, 2, 7, 16):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_passthrough():
    from madm_amd import dist as mdist
    os.environ.pop("WORLD_SIZE", None)
    assert mdist.max_over_ranks(3.5) == 3.5 and mdist.sum_over_ranks(2) == 2.0


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from madm_amd import dist as mdist
    d = mdist.init(backend="gloo")
    n = 1000
    out = {}
    # fp32 all-reduce (DDP's path), reduce-scatter + all-gather (the mesh-friendly form), and both with a 16-bit wire
    for tag, kw in (("allreduce", {}), ("rs_ag", dict(mode="rs_ag")), ("bf16", dict(wire_dtype=torch.bfloat16)),
                    ("rs_ag_f16", dict(mode="rs_ag", wire_dtype=torch.float16))):
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)      # rank r holds (r + 1) * [0, 1, 2, ...]
        red = mdist.GradBucketReducer(g, d, bucket_numel=97, **kw)  # ragged AND odd: 1000 = 10 * 97 + 30, tails not / world
        red.reduce_tail(701)                                       # the explicit backward fills the buffer back to front
        red.reduce_tail(250)
        red.finish()
        out[tag] = g.numpy().copy()
    q.put((rank, out))
    d.barrier()
    d.destroy_process_group()


def test_gradient_all_reduce_mean_two_rank_gloo():
    """The training path's one exchange step: bucketed all-reduce(mean) of the flat gradient buffer, handed over in
    back-to-front spans; every rank ends with the mean of the ranks' gradients, every element reduced exactly once."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((r, g) for r, g in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.arange(1000, dtype=torch.float32) * 1.5           # mean of 1x and 2x
    for _, out in res:
        assert torch.equal(torch.from_numpy(out["allreduce"]), want)          # fp32 modes: exact, every element once
        assert torch.equal(torch.from_numpy(out["rs_ag"]), want)
        # 16-bit wire (DDP's fp16 / bf16 compress hooks): each addend (g / world) and the sum are rounded to the wire type ->
        # relative error <= 2^-8 (bf16: 8 significand bits) / 2^-11 (f16) per rounding, two roundings; stated tolerance 1.5x that
        for tag, eps in (("bf16", 2.0 ** -8), ("rs_ag_f16", 2.0 ** -11)):
            got = torch.from_numpy(out[tag])
            assert ((got - want).abs() <= 1.5 * 2 * eps * want.abs()).all(), tag
            assert not torch.equal(got, want)                                  # (it IS lossy: 1.5 * 999 is not a bf16 / f16 number)
    assert all(np.array_equal(res[0][1][k], res[1][1][k]) for k in res[0][1])  # the ranks agree bit for bit in every mode


def test_gradient_reducer_is_a_no_op_for_one_rank():
    from madm_amd.dist import GradBucketReducer
    g = torch.arange(10, dtype=torch.float32)
    red = GradBucketReducer(g, None)
    red.reduce_tail(4)
    red.finish()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))


class _TinyTrainModel(torch.nn.Module):
    """Stand-in with MTMADISE's parameter-name families (what MadmTrainer's flat-buffer ordering keys on)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g))
        self.backbone = torch.nn.Module()
        self.backbone.clip_project_rgb = torch.nn.ParameterDict({"prompt_embed": mk(7, 5)})
        unet = torch.nn.Module()
        unet.conv_in = torch.nn.ParameterDict({"weight": mk(2000)})
        unet.time_embedding = torch.nn.ParameterDict({"weight": mk(300)})
        unet.res = torch.nn.Module()
        unet.res.time_emb_proj = torch.nn.ParameterDict({"weight": mk(500)})
        unet.res.conv1 = torch.nn.ParameterDict({"weight": mk(3000)})
        unet.up = torch.nn.ParameterDict({"weight": mk(2500)})
        self.backbone.unet = unet
        self.sem_seg_head = torch.nn.ParameterDict({"weight": mk(1500), "bias": mk(11)})
        # the EMA teacher: requires_grad=False parameters (cmdise.py:307-335) that DDP's start-up broadcast covers too
        self.ema_sem_seg_head = torch.nn.ParameterDict({"weight": mk(1500), "bias": mk(11)})
        for p in self.ema_sem_seg_head.parameters():
            p.requires_grad = False


def _trainer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from madm_amd import dist as mdist
    from madm_amd.train import MadmTrainer
    d = mdist.init(backend="gloo")
    model = _TinyTrainModel()
    if rank == 1:                                   # ranks start from DIFFERENT parameters: rank 0's are broadcast
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    tr = MadmTrainer(model, lr=1e-3, weight_decay=0.05, dist=d)
    tr.reducer.bucket = 2048                        # small buckets so that several collectives start during the "backward"
    params0 = torch.cat([tr.opt.flat.flat.clone()] + [p.detach().flatten() for p in model.ema_sem_seg_head.parameters()])
    order = [n for n, _ in model.named_parameters()]
    flat_names = [dict((id(p), n) for n, p in model.named_parameters())[id(p)] for p in tr.opt.flat.params]
    # the explicit backward finishes gradients back to front: head, up, res.conv1, conv_in, then the late tails
    tr.opt.zero_grad()
    seq = ["sem_seg_head.weight", "sem_seg_head.bias", "backbone.unet.up.weight", "backbone.unet.res.conv1.weight",
           "backbone.unet.conv_in.weight", "backbone.unet.res.time_emb_proj.weight", "backbone.unet.time_embedding.weight",
           "backbone.clip_project_rgb.prompt_embed"]
    named = dict(model.named_parameters())
    for n in seq:
        p = named[n]
        tr.final(p, torch.full(p.shape, float(rank + 1)) * (1 + len(n)))
    during = tr.reduced_during_backward
    tr.reducer.finish()
    # numpy arrays travel by value (torch tensors go through shared-memory handles that die with the worker)
    q.put((rank, params0.numpy(), tr.opt.flat.grad.numpy().copy(), flat_names, during, [tr.opt.flat.offsets[i] for i in range(len(seq))]))
    d.barrier()
    d.destroy_process_group()


def test_trainer_broadcast_flat_order_and_overlapped_reduce_two_rank_gloo():
    """MadmTrainer's DDP contract on CPU / gloo (no kernels involved): rank 0's parameters are broadcast at construction
    (main.py:289-294), the flat buffer puts the late-finishing gradients first, finished tails are all-reduced in whole
    buckets WHILE the remaining gradients still arrive, and every rank ends with the mean gradient."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, names0, during0, _), (_, p1, g1, names1, during1, _) = res
    p0, g0, p1, g1 = (torch.from_numpy(a) for a in (p0, g0, p1, g1))
    assert torch.equal(p0, p1)                                        # broadcast: identical start, frozen teacher included
    ref_teacher = torch.cat([p.detach().flatten() for p in _TinyTrainModel().ema_sem_seg_head.parameters()])
    assert torch.equal(p1[-ref_teacher.numel():], ref_teacher)        # ... and it is rank 0's (unperturbed) teacher
    assert names0 == names1 and names0[:3] == ["backbone.clip_project_rgb.prompt_embed", "backbone.unet.time_embedding.weight",
                                                "backbone.unet.res.time_emb_proj.weight"]
    assert names0[3:] == ["backbone.unet.conv_in.weight", "backbone.unet.res.conv1.weight", "backbone.unet.up.weight",
                          "sem_seg_head.bias", "sem_seg_head.weight"]      # (ParameterDict sorts its keys)
    assert torch.equal(g0, g1)                                        # every rank holds the same (mean) gradient
    assert during0 == during1 and during0 >= 2 * 2048                 # buckets were handed over before the backward ended
    m = _TinyTrainModel()
    tot = 0
    named = dict(m.named_parameters())
    # mean over the two ranks of (rank + 1) * (1 + len(name)) = 1.5 * (1 + len(name)), padding stays zero
    from madm_amd import optim
    for n in names0:
        numel = named[n].numel()
        assert torch.all(g0[tot:tot + numel] == 1.5 * (1 + len(n))), n
        pad = (numel + 1023) // 1024 * 1024
        assert torch.all(g0[tot + numel:tot + pad] == 0)
        tot += pad
