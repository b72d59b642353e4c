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


def scatter_assignments(lengths: Sequence[int], rank: int, world: int, device="cpu"):
    """Rank 0 computes the partition and scatters one (begin, end) pair to every rank."""
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
