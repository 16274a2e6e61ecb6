"""N > 1 path on CPU: two gloo ranks exercise the stream sharding and the reporting reductions
that bench.py uses with RCCL on the GPUs (no data-path collective exists)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdrdaemon_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sharding.stream_ids(rank, world, 8)
    # every rank "processes" its streams: elapsed differs per rank, samples = 8 streams x 1000
    elapsed, total = sharding.aggregate(0.5 + 0.25 * rank, len(ids) * 1000, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, ids)
    q.put((rank, ids, elapsed, total, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    all_ids = res[0][1] + res[1][1]
    assert all_ids == list(range(16))  # disjoint cover: stream s on rank s // 8
    for rank, ids, elapsed, total, gathered in res:
        assert all(sharding.owner(i, 8) == rank for i in ids)
        assert elapsed == pytest.approx(0.75)  # MAX over ranks
        assert total == 16000.0  # SUM over ranks
        assert gathered == [res[0][1], res[1][1]]


def test_single_process_passthrough():
    assert sharding.aggregate(1.5, 10) == (1.5, 10.0)
    assert sharding.stream_ids(3, 8, 8) == list(range(24, 32))
    with pytest.raises(ValueError):
        sharding.stream_ids(8, 8, 8)
