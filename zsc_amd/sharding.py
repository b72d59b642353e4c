"""Multi-GPU sharding of a batch of independent buffers.

Every buffer is one independent zsc_compress call, so a node's GPUs never exchange
payload: the list of buffers is cut into contiguous ranges balanced by input bytes,
rank 0 scatters the *assignment table* (which range each rank owns) and gathers the
per-buffer result sizes.  Both are a few kilobytes of metadata, so they go through
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in
the CPU tests); there is no data-path collective.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def partition_by_bytes(lengths: Sequence[int], parts: int) -> List[Tuple[int, int]]:
    """Cut range(len(lengths)) into `parts` contiguous [begin, end) ranges whose byte
    totals are as even as a contiguous cut allows (greedy on the running prefix sum)."""
    n = len(lengths)
    total = sum(lengths)
    out, begin, acc = [], 0, 0
    for p in range(parts):
        if p == parts - 1:
            out.append((begin, n))
            break
        target = total * (p + 1) / parts
        end = begin
        while end < n - (parts - 1 - p) and acc + lengths[end] / 2 <= target:
            acc += lengths[end]
            end += 1
        out.append((begin, end))
        begin = end
    return out


# Relative cost of one input byte per class of content, from the parse kernel's measured rates on
# one MI355X (text 3.1, incompressible 3.1, table-like 8.9, run-heavy /
# bitmap 3.7, all-zero ~9 GB/s): what a strong-scaling partition (a FIXED batch cut over the GPUs,
# BASELINE config 5) balances instead of plain bytes.  Unknown classes cost like text.
CLASS_COST = {"text": 1.0, "token": 1.0, "object": 1.0, "random": 1.0, "table": 0.35, "bitmap": 0.85,
              "runs": 0.85, "zero": 0.35}


def cost_proxy(lengths: Sequence[int], kinds: Sequence[str]) -> List[float]:
    """Estimated relative parse cost of every buffer: bytes x CLASS_COST[kind]."""
    return [n * CLASS_COST.get(k, 1.0) for n, k in zip(lengths, kinds)]


def scatter_assignments(lengths: Sequence[int], rank: int, world: int, device="cpu"):
    """Rank 0 computes the partition and scatters one (begin, end) pair to every rank.
    `lengths` may be any per-buffer weight (bytes for weak scaling, cost_proxy() for a fixed batch)."""
    import torch
    import torch.distributed as dist

    mine = torch.zeros(2, dtype=torch.int64, device=device)
    if world == 1:
        return (0, len(lengths))
    if rank == 0:
        parts = partition_by_bytes(lengths, world)
        table = [torch.tensor(p, dtype=torch.int64, device=device) for p in parts]
        dist.scatter(mine, scatter_list=table, src=0)
    else:
        dist.scatter(mine, scatter_list=None, src=0)
    return int(mine[0]), int(mine[1])


def gather_sizes(local_sizes: Sequence[int], total: int, begin: int, rank: int, world: int,
                 device="cpu") -> List[int]:
    """Every rank contributes the compressed sizes of its range; all ranks get the full list."""
    import torch
    import torch.distributed as dist

    full = torch.zeros(total, dtype=torch.int64, device=device)
    full[begin:begin + len(local_sizes)] = torch.tensor(list(local_sizes), dtype=torch.int64,
                                                        device=device)
    if world > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return [int(x) for x in full.cpu()]


# ---- root-staged variant (SURVEY 8e "north-star variant", BASELINE config 5) ------------------
#
# The whole batch lies in GPU 0's memory; every other rank receives ITS byte range of it and sends
# its streams back.  Point-to-point from / to the root (on the GPU box: one ncclGroup of
# ncclSend / ncclRecv, which uses all of the root's xGMI links at once -- not a ring), payload as
# plain bytes, ranges cut where buffers start.  Only bench.py --stage-on-root and the gloo test use
# this; the default path never moves payload between GPUs.

def scatter_payload(full, byte_ranges: Sequence[Tuple[int, int]], rank: int, world: int, device="cpu"):
    """`full`: the staged batch on rank 0 (uint8 tensor; None elsewhere).  Returns this rank's
    slice [byte_ranges[rank]) as a tensor on `device`."""
    import torch
    import torch.distributed as dist

    b, e = byte_ranges[rank]
    if world == 1:
        return full[b:e]
    mine = torch.empty(e - b, dtype=torch.uint8, device=device)
    if rank == 0:
        mine.copy_(full[b:e])
        ops = [dist.P2POp(dist.isend, full[rb:re].contiguous(), r)
               for r, (rb, re) in enumerate(byte_ranges) if r != 0 and re > rb]
    else:
        ops = [dist.P2POp(dist.irecv, mine, 0)] if e > b else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return mine


def gather_payload(local, byte_ranges: Sequence[Tuple[int, int]], rank: int, world: int, device="cpu"):
    """The reverse: every rank's `local` bytes (its streams, laid out back to back as the root
    expects them) arrive at byte_ranges[rank] of one tensor on rank 0, which is returned there
    (None elsewhere).  len(local) must equal the rank's range."""
    import torch
    import torch.distributed as dist

    b, e = byte_ranges[rank]
    assert local.numel() == e - b, (local.numel(), b, e)
    if world == 1:
        return local
    full = None
    if rank == 0:
        full = torch.empty(max(re for _, re in byte_ranges), dtype=torch.uint8, device=device)
        full[b:e] = local
        ops = [dist.P2POp(dist.irecv, full[rb:re], r)
               for r, (rb, re) in enumerate(byte_ranges) if r != 0 and re > rb]
    else:
        ops = [dist.P2POp(dist.isend, local.contiguous(), 0)] if e > b else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return full
