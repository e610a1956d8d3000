"""N > 1 path on CPU: stream sharding + the statistics reduction over gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zignal_amd import dist as zdist


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 1 << 20, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            edges = [zdist.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = zdist.shard_range(1000, rank, world)
        # rank-local "work": seconds differ per rank, samples = shard size * T, checksum = sum of ids
        st = zdist.reduce_stats(seconds=1.0 + rank, samples=float((e - b) * 16), checksum=sum(range(b, e)))
        q.put((rank, st))
    finally:
        dist.destroy_process_group()


def test_reduce_stats_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["world"] == 2
        assert res[r]["seconds"] == 2.0                      # max over ranks
        assert res[r]["samples"] == 1000 * 16                # sum over ranks
        assert res[r]["checksum"] == sum(range(1000)) and isinstance(res[r]["checksum"], int)


def test_reduce_stats_single_process_identity():
    st = zdist.reduce_stats(0.5, 10.0, 3)
    assert st == {"seconds": 0.5, "samples": 10.0, "checksum": 3, "world": 1}
