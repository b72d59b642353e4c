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


def _payload_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle_py import Oracle
    from zsc_amd import corpus
    oracle = Oracle()
    lens = [3000, 500, 12000, 7, 0, 900, 4000, 2500, 16, 5000]
    # the batch layout of the plans: every buffer starts 16-byte aligned
    offs, o = [], 0
    for n in lens:
        offs.append(o)
        o += (n + 15) & ~15
    total = o
    full = None
    bufs = [corpus.make_buffer(("text", "table")[i % 2], n, 70 + i) for i, n in enumerate(lens)]
    if rank == 0:  # only the root has the payload
        full = torch.zeros(total, dtype=torch.uint8)
        for off, b in zip(offs, bufs):
            if b:
                full[off:off + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    begin, end = sharding.scatter_assignments(lens, rank, world)
    cuts = sharding.partition_by_bytes(lens, world)
    in_ranges = [(offs[b] if b < len(lens) else total, offs[e] if e < len(lens) else total) for b, e in cuts]
    mine = sharding.scatter_payload(full, in_ranges, rank, world)
    base = in_ranges[rank][0]
    got_in = [bytes(mine[offs[i] - base:offs[i] - base + lens[i]].numpy()) for i in range(begin, end)]
    ok_in = got_in == bufs[begin:end] if rank == 0 else all(len(x) == lens[begin + k] for k, x in enumerate(got_in))
    streams = [oracle.compress(b, 6)[1] for b in got_in]  # (the GPU path on the box)
    sizes = sharding.gather_sizes([len(x) for x in streams], len(lens), begin, rank, world)
    ooffs, o = [], 0
    for n in sizes:
        ooffs.append(o)
        o += n
    ooffs.append(o)
    out_ranges = [(ooffs[b], ooffs[e]) for b, e in cuts]
    local = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8) if streams and sum(map(len, streams)) else \
        torch.zeros(0, dtype=torch.uint8)
    allout = sharding.gather_payload(local, out_ranges, rank, world)
    ok_out = True
    if rank == 0:
        want = b"".join(oracle.compress(b, 6)[1] for b in bufs)
        ok_out = bytes(allout.numpy()) == want
    q.put((rank, bool(ok_in), bool(ok_out)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_root_staged_payload_gloo():
    """BASELINE config 5's wording: the batch staged on rank 0, payload scattered point to point,
    streams gathered back (zsc_amd.sharding.scatter_payload / gather_payload)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_payload_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g[1] and g[2] for g in got), got
