"""world_size-2 gloo test of the multi-GPU plumbing (madm_amd/dist.py): shards cover the batch exactly
once, the barrier + MAX reduction of the timed region and the SUM of processed units behave as bench.py
assumes.  Runs on CPU."""
import os
import socket

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
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)          # rank r holds (r + 1) * [0, 1, 2, ...]
    red = mdist.GradBucketReducer(g, d, bucket_numel=96)           # ragged: 1000 = 10 * 96 + 40
    red.reduce_tail(700)                                           # the explicit backward fills the buffer back to front
    red.reduce_tail(250)
    red.finish()
    q.put((rank, g.clone()))
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
    for _, g in res:
        assert torch.equal(g, want)


def test_gradient_reducer_is_a_no_op_for_one_rank():
    from madm_amd.dist import GradBucketReducer
    g = torch.arange(10, dtype=torch.float32)
    red = GradBucketReducer(g, None)
    red.reduce_tail(4)
    red.finish()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))
