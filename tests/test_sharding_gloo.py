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


def test_cost_proxy_moves_the_cut_towards_the_cheap_class():
    # twelve 64 KiB buffers: the first six all-zero (cheap), the last six text (dear)
    lens, kinds = [65536] * 12, ["zero"] * 6 + ["text"] * 6
    by_bytes = sharding.partition_by_bytes(lens, 2)
    by_cost = sharding.partition_by_bytes(sharding.cost_proxy(lens, kinds), 2)
    assert by_bytes == [(0, 6), (6, 12)]
    assert by_cost[0][1] > 6 and by_cost[1][1] == 12   # the first rank also takes some of the dear ones


def _worker_config5(rank, world, port, q):
    """bench.py --config5 at toy size: a fixed batch of fixed-stride buffers staged on rank 0, cut by the
    cost proxy, scattered point to point, "compressed" (the oracle stands in for the device), streams
    gathered back to rank 0 at their fixed stride."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle_py import Oracle
    from zsc_amd import corpus
    oracle = Oracle()
    n, count, stride_in, stride_out = 2048, 9, 2048 + 64, 2304
    kinds = ["random", "zero", "text"] * 3
    lens = [n] * count
    bufs = [corpus.make_buffer(k, n, 70 + i) for i, k in enumerate(kinds)]
    ub, ue = sharding.scatter_assignments(sharding.cost_proxy(lens, kinds), rank, world)
    tbl = torch.zeros(2 * world, dtype=torch.int64)
    tbl[2 * rank], tbl[2 * rank + 1] = ub, ue
    dist.all_reduce(tbl)
    ranges = [(int(tbl[2 * r]), int(tbl[2 * r + 1])) for r in range(world)]
    full = None
    if rank == 0:
        full = torch.zeros(count * stride_in, dtype=torch.uint8)
        for i, b in enumerate(bufs):
            full[i * stride_in:i * stride_in + n] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    mine = sharding.scatter_payload(full, [(b * stride_in, e * stride_in) for b, e in ranges], rank, world)
    out = torch.zeros((ue - ub) * stride_out, dtype=torch.uint8)
    sizes = []
    for k in range(ue - ub):
        src = bytes(mine[k * stride_in:k * stride_in + n].numpy())
        z = oracle.compress(src, 6)[1]
        out[k * stride_out:k * stride_out + len(z)] = torch.frombuffer(bytearray(z), dtype=torch.uint8)
        sizes.append(len(z))
    got = sharding.gather_payload(out, [(b * stride_out, e * stride_out) for b, e in ranges], rank, world)
    all_sizes = sharding.gather_sizes(sizes, count, ub, rank, world)
    ok = True
    if rank == 0:
        for i, b in enumerate(bufs):
            z = oracle.compress(b, 6)[1]
            ok = ok and all_sizes[i] == len(z) and bytes(got[i * stride_out:i * stride_out + len(z)].numpy()) == z
    q.put((rank, ranges, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_config5_flow_two_ranks_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_config5, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = got[0][1]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 9
    assert all(g[2] for g in got)
