"""The N>1 path of bench.py / zsc_amd.sharding on CPU: two gloo ranks scatter the
assignment table, "compress" their shard (sizes from the oracle) and gather sizes."""
import os
import socket

import pytest

from zsc_amd import sharding


def test_partition_by_bytes_is_contiguous_and_balanced():
    lens = [152089, 513216, 11150, 1029744, 38240, 426754, 481861, 24603, 3721, 4227, 125179] * 16
    for parts in (1, 2, 3, 4, 8):
        cuts = sharding.partition_by_bytes(lens, parts)
        assert cuts[0][0] == 0 and cuts[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sums = [sum(lens[b:e]) for b, e in cuts]
        assert max(sums) - min(sums) <= 2 * max(lens)
    assert sharding.partition_by_bytes([5], 4)[-1][1] == 1
    assert sharding.partition_by_bytes([], 2) == [(0, 0), (0, 0)]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle_py import Oracle
    from zsc_amd import corpus
    oracle = Oracle()
    lens = [3000, 500, 12000, 7, 0, 900, 4000, 2500]
    bufs = [corpus.make_buffer("text", n, 40 + i) for i, n in enumerate(lens)]
    begin, end = sharding.scatter_assignments(lens, rank, world)
    local = [len(oracle.compress(b, 6)[1]) for b in bufs[begin:end]]
    sizes = sharding.gather_sizes(local, len(lens), begin, rank, world)
    want = [len(oracle.compress(b, 6)[1]) for b in bufs]
    q.put((rank, begin, end, sizes == want))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_scatter_gather_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == 8
    assert all(g[3] for g in got)
