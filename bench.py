#!/usr/bin/env python3
"""bench.py -- deflate level-6 throughput of the MI355X path on Canterbury-like x N.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one full pass of the hot path (checksum, hash sort, LZ77 parse, Huffman
plan, layout, bit packing) over one batch that is already resident in HBM.  The
batch is BASELINE.json config 2: the 11-buffer Canterbury-like set (real Canterbury
file sizes, synthetic seeded contents -- the corpus is not available offline) x 4096
independent buffers per GPU, level 6, one zsc_compress call per buffer.  With N GPUs
every rank compresses its own x4096 batch (weak scaling); buffers never cross GPUs,
only the assignment table and the result sizes travel (RCCL, metadata only).

Rank 0 prints ONE JSON line:  value = uncompressed MB/s in, whole job.
  roofline      the dominant kernel (LZ77 parse): algorithmic bytes (input read once +
                stream written once) / its mean duration measured with HIP
                events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  the same workload on the host cores: the compiled reference when
                oracle/_ref/libzsc_ref.so travelled with the repo ("reference"),
                otherwise the byte-identical oracle ("port"); bounded sample.
  checked       every distinct buffer of the timed batch against the oracle, and every
                replica against its first copy on the device.
With one GPU the line also carries the rest of BASELINE's metric (untimed by the headline):
  inflate       BASELINE config 4: gzip members of 4-64 KiB (made by this library's own
                deflate, checked against the oracle) replicated to --inflate-streams, MB/s
                out with its own roofline (k_inflate) and CPU baseline.
  levels_64k    BASELINE config 3: 1 GiB of 64 KiB random / zero / text buffers at
                levels 1, 6 and 9.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(level: int, seconds_budget: float = 20.0):
    """Time the CPU codec on the host cores over whole Canterbury-like sets (bounded sample)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle_py import Oracle, Reference
    from zsc_amd import corpus

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a one-GPU box owns a 16-core share of the host (more threads would only oversubscribe it)
    nproc = cores
    cores = max(1, min(cores, int(os.environ.get("ZSC_BENCH_CPU_THREADS", "16"))))
    kind = "reference" if Reference.available() else "port"
    sets = [corpus.canterbury_like(s) for s in range(2)]
    set_bytes = sum(len(b) for _, b in sets[0])

    def worker(idx: int):
        codec = Reference() if kind == "reference" else Oracle()  # private work buffer per thread
        done = 0
        t_end = time.perf_counter() + seconds_budget
        rounds = 0
        while True:
            for _, b in sets[(idx + rounds) % len(sets)]:
                res = codec.compress(b, level)
                assert res[0] == 0
                done += len(b)
            rounds += 1
            if time.perf_counter() >= t_end or rounds >= 64:
                return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(total / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "host_cpus": nproc, "kind": kind,
            "sample": f"{total // set_bytes} Canterbury-like sets ({total / 1e6:.0f} MB) at level "
                      f"{level}, one codec instance per thread, {dt:.1f} s wall"}


def cpu_baseline_bufs(bufs, level: int, seconds_budget: float):
    """The CPU codec on the host cores over a given list of buffers (bounded sample), as cpu_baseline."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle_py import Oracle, Reference

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    nproc = cores
    cores = max(1, min(cores, int(os.environ.get("ZSC_BENCH_CPU_THREADS", "16"))))
    kind = "reference" if Reference.available() else "port"

    def worker(idx: int):
        codec = Reference() if kind == "reference" else Oracle()
        done, k = 0, idx
        t_end = time.perf_counter() + seconds_budget
        while time.perf_counter() < t_end:
            b = bufs[k % len(bufs)]
            assert codec.compress(b, level)[0] == 0
            done += len(b)
            k += cores
        return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(total / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "host_cpus": nproc, "kind": kind,
            "sample": f"{total / 1e6:.0f} MB from the {len(bufs)} distinct buffers at level {level}, one codec "
                      f"instance per thread, {dt:.1f} s wall"}


def cpu_inflate_baseline(streams, outs, seconds_budget: float = 10.0):
    """Time the CPU decoder on the host cores over the distinct gzip members (bounded sample)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle_py import Oracle, Reference

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, int(os.environ.get("ZSC_BENCH_CPU_THREADS", "16"))))
    kind = "reference" if Reference.available() else "port"

    def worker(idx: int):
        codec = Reference() if kind == "reference" else Oracle()
        done = 0
        t_end = time.perf_counter() + seconds_budget
        k = idx
        while time.perf_counter() < t_end:
            s, o = streams[k % len(streams)], outs[k % len(streams)]
            res = codec.uncompress(s, len(o), window_bits=31)
            assert res[0] == 0 and len(res[1]) == len(o)
            done += len(o)
            k += cores
        return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(total / dt / 1e6, 2), "unit": "MB/s out", "cores": cores, "kind": kind,
            "sample": f"{total / 1e6:.0f} MB of output from the {len(streams)} distinct gzip members, "
                      f"zsc_uncompress_gzip semantics, one codec instance per thread, {dt:.1f} s wall"}


def bench_inflate(dev, stream, nstreams: int, distinct: int, fence, orders=("neighbours_differ", "neighbours_identical")):
    """BASELINE config 4: inflate-only over gzip members of 4-64 KiB."""
    import torch
    import zsc_amd
    from zsc_amd import corpus
    from oracle.oracle_py import Oracle

    distinct = max(1, min(distinct, nstreams))
    st = corpus.Stream(4242, 3)
    sizes = [4096 + int(x) for x in st.below(distinct, 65536 - 4096 + 1)]
    kinds = ("text", "text", "token", "table")
    bufs = [corpus.make_buffer(kinds[i % 4], sizes[i], 7000 + i) for i in range(distinct)]
    # the members: this library's own deflate (gzip wrapper), every one checked against the oracle
    rc, members, stats = zsc_amd.compress_batch(bufs, level=6, window_bits=31)
    if rc != 0 or any(x != 0 for x in stats):
        raise SystemExit("inflate bench: making the gzip members failed")
    oracle = Oracle()
    nchk = min(distinct, int(os.environ.get("ZSC_BENCH_INFLATE_CHECK", "256")))
    for i in range(nchk):
        if oracle.compress(bufs[i], 6, window_bits=31)[1] != members[i]:
            raise SystemExit(f"inflate bench: gzip member {i} differs from the oracle")
    slens = [len(m) for m in members]
    reps = (nstreams + distinct - 1) // distinct
    all_slens = (slens * reps)[:nstreams]
    all_caps = (sizes * reps)[:nstreams]
    # Two orders of the same batch.  The plan decodes streams longest first, four to a wavefront,
    # so the 512 replicas of a member would sit next to each other and the four streams of a
    # wavefront would be identical (they never diverge: the best case).  With an explicit decode
    # order neighbours are DIFFERENT members of nearly the same length -- what a batch of
    # all-different streams looks like -- and that is the figure reported as `value`.
    results = {}
    for mode in orders:
        order = None
        if mode == "neighbours_differ" and nstreams % distinct == 0 and nstreams > distinct:
            by_len = sorted(range(nstreams), key=lambda i: -all_caps[i])  # (stable: the library's own order)
            r = nstreams // distinct
            order = [by_len[(k % distinct) * r + k // distinct] for k in range(nstreams)]
        ip = zsc_amd.InflatePlan(all_slens, all_caps, window_bits=31, decode_order=order)
        # one period of the source layout on the host, replicated on the device
        per_src = ip.src_offsets[distinct] if nstreams > distinct else ip.src_bytes - 64
        per_dst = ip.dst_offsets[distinct] if nstreams > distinct else ip.dst_bytes - 64
        host = torch.zeros(per_src, dtype=torch.uint8)
        for off, m in zip(ip.src_offsets, members):
            host[off:off + len(m)] = torch.frombuffer(bytearray(m), dtype=torch.uint8)
        d_per = host.to(dev)
        d_src = torch.zeros(ip.src_bytes, dtype=torch.uint8, device=dev)
        d_src[:ip.src_bytes - 64] = d_per.repeat(reps)[:ip.src_bytes - 64]
        d_dst = torch.empty(ip.dst_bytes, dtype=torch.uint8, device=dev)
        ip.run(d_src.data_ptr(), d_dst.data_ptr(), stream)
        ip.results()
        fence()
        t1 = time.perf_counter()
        steps = 2
        kms = 0.0
        for _ in range(steps):
            ip.run(d_src.data_ptr(), d_dst.data_ptr(), stream)
            kms += ip.results()[3]
        fence()
        wall = (time.perf_counter() - t1) / steps
        kms /= steps
        olens, used, istat, _ = ip.results()
        ok = all(x == 0 for x in istat) and olens == all_caps and used == all_slens
        # every distinct member's bytes against its source; every replica against the first copy
        first = d_dst[:per_dst].cpu()
        for i in range(distinct):
            got = bytes(first[ip.dst_offsets[i]:ip.dst_offsets[i] + sizes[i]].numpy())
            ok = ok and got == bufs[i]
        mask = torch.zeros(per_dst, dtype=torch.bool)
        for i in range(distinct):
            mask[ip.dst_offsets[i]:ip.dst_offsets[i] + sizes[i]] = True
        d_mask = mask.to(dev)
        full = (ip.dst_bytes - 64) // per_dst
        rows = d_dst[:full * per_dst].view(full, per_dst)
        ok = ok and bool((rows[:, d_mask] == rows[0, d_mask]).all())
        results[mode] = (wall, kms, bool(ok))
        ip.close()
        del d_src, d_dst, rows, d_mask, first
    wall, kms, ok = results[orders[0]]
    wall2, kms2, ok2 = results[orders[-1]]
    ok = ok and ok2
    out_bytes, in_bytes = sum(all_caps), sum(all_slens)
    achieved = (out_bytes + in_bytes) / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    info = {"metric": "inflate uncompressed MB/s out, gzip members of 4-64 KiB (BASELINE config 4)",
            "value": round(out_bytes / wall / 1e6, 2), "unit": "MB/s", "ms_per_step": round(wall * 1e3, 3),
            "streams": nstreams, "distinct_members": distinct, "output_bytes": out_bytes,
            "compressed_bytes": in_bytes, "all_ok": bool(ok),
            "roofline": {"bound": "hbm", "kernel": "k_inflate", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "kernel_ms": round(kms, 3), "algorithmic_bytes_per_launch": out_bytes + in_bytes,
                         "traffic": None},
            "order": orders[0],
            "identical_neighbours": {"value": round(out_bytes / wall2 / 1e6, 2), "kernel_ms": round(kms2, 3),
                                     "note": "the same batch in plain length order: the replicas of a member share "
                                             "wavefronts and never diverge -- an artefact of replication, not a "
                                             "property of a real batch"},
            "note": "zsc_uncompress_gzip semantics (header, CRC-32 and ISIZE checked), one 16-lane group per "
                    "member (four members per wavefront, taken from one queue), members and outputs resident in "
                    "HBM; all outputs compared with their sources; `value`: neighbours in the decode order are "
                    "different members (an explicit decode order, zsc_hip_inflate_plan_create_ordered)"}
    return info, members, bufs


def bench_levels_64k(dev, stream, count: int, fence, cpu: bool = True):
    """BASELINE config 3: count x 64 KiB (random / zero / text) at levels 1, 6, 9."""
    import torch
    import zsc_amd
    from zsc_amd import corpus
    from oracle.oracle_py import Oracle

    distinct = min(count, 96)
    bufs = corpus.mix64k(distinct, 5)
    oracle = Oracle()
    out = {}
    for level in (1, 6, 9):
        plan = zsc_amd.DeflatePlan([65536] * count, level=level)
        per = plan.in_offsets[distinct] if count > distinct else plan.in_bytes - 64
        host = torch.zeros(per, dtype=torch.uint8)
        for off, b in zip(plan.in_offsets, bufs):
            host[off:off + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        reps = (count + distinct - 1) // distinct
        d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device=dev)
        d_in[:plan.in_bytes - 64] = host.to(dev).repeat(reps)[:plan.in_bytes - 64]
        d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
        plan.run(d_in.data_ptr(), d_out.data_ptr(), stream)
        plan.results()
        fence()
        plan.profile(True)
        t1 = time.perf_counter()
        steps = 2
        for _ in range(steps):
            plan.run(d_in.data_ptr(), d_out.data_ptr(), stream)
        fence()
        wall = (time.perf_counter() - t1) / steps
        lens, stats = plan.results()
        kt = plan.kernel_times_ms()
        ok = all(x == 0 for x in stats)
        head = d_out[:plan.out_offsets[distinct - 1] + plan.out_caps[distinct - 1]].cpu()
        for k in range(distinct):  # every distinct buffer against the oracle
            got = bytes(head[plan.out_offsets[k]:plan.out_offsets[k] + lens[k]].numpy())
            ok = ok and got == oracle.compress(bufs[k], level)[1]
        ok = ok and all(lens[i] == lens[i % distinct] for i in range(count))
        alg = 65536 * count + sum(lens)
        pk = "parse"
        achieved = alg / (kt[pk] * 1e-3) / 1e9 if kt[pk] > 0 else 0.0
        out[f"L{level}"] = {"MB_per_s_in": round(65536 * count / wall / 1e6, 2), "ms_per_step": round(wall * 1e3, 3),
                            "compressed_bytes": sum(lens), "all_ok": bool(ok),
                            "roofline": {"bound": "hbm", "kernel": "k_parse_fast" if level < 4 else "k_parse_seg",
                                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(achieved / HBM_PEAK_GBS, 6), "kernel_ms": round(kt[pk], 3),
                                         "algorithmic_bytes_per_launch": alg, "traffic": None},
                            "kernel_ms": {k: round(v, 3) for k, v in kt.items()}}
        if cpu:
            out[f"L{level}"]["cpu_baseline"] = cpu_baseline_bufs(bufs, level, 4.0)
        plan.close()
        del d_in, d_out
    return {"metric": "deflate uncompressed MB/s in, 64 KiB random / zero / text buffers (BASELINE config 3)",
            "buffers": count, "input_bytes": 65536 * count, "levels": out,
            "note": f"{distinct} distinct buffers replicated; every distinct stream compared with the oracle"}


# the reference's own published totals for the Canterbury corpus (2 810 784 B) compressed by
# zsc_compress with max_block_len 100 000: /root/reference/README.md:170-179, Raspberry Pi 4
README_TOTALS = {0: 2811285, 1: 842979, 2: 821886, 3: 809844, 4: 768442, 5: 747916, 6: 744167, 7: 742155,
                 8: 738951, 9: 738674}


def bench_corpus_dir(path: str):
    """if the REAL Canterbury files are there, compress them as the reference's performance
    test does (zsc_compress, max_block_len 100 000, test/zlib_gtest.cpp:2406) and put the totals next to
    the reference's published ones.  Host pointers through zsc_hip_compress_sections_batch; untimed."""
    import zsc_amd
    from zsc_amd import corpus
    from oracle.oracle_py import Oracle

    names = [n for n, _, _ in corpus.CANTERBURY_LIKE]
    found = [n for n in names if os.path.isfile(os.path.join(path, n))]
    info = {"dir": path, "files_found": len(found), "of": len(names)}
    if not found:
        info["note"] = "none of the 11 Canterbury files is in this directory"
        return info
    bufs = [open(os.path.join(path, n), "rb").read() for n in found]
    info["input_bytes"] = sum(len(b) for b in bufs)
    oracle = Oracle()
    levels = {}
    for level in (1, 6, 9):
        rc, outs, stats = zsc_amd.compress_sections_batch(bufs, [100000] * len(bufs), level=level)
        ok = rc == 0 and all(x == 0 for x in stats)
        for b, o in zip(bufs, outs):  # the checker: the oracle's call-by-call restatement of the wrapper
            cap = zsc_amd.compress_get_max_output_size2(len(b), 100000, level)[1]
            want = oracle.compress(b, level, max_block_len=100000, dest_cap=cap)
            ok = ok and (want[0], want[1]) == (0, o)
        total = sum(len(o) for o in outs)
        levels[f"L{level}"] = {"compressed_bytes": total, "reference_README_total": README_TOTALS[level],
                               "equal": total == README_TOTALS[level] if len(found) == len(names) else None,
                               "streams_equal_oracle": bool(ok)}
    info["levels"] = levels
    info["note"] = ("the reference publishes totals for the whole corpus only; `equal` is null when files are "
                    "missing")
    return info


def run_config5(args, rank, world, local, dev):
    """BASELINE config 5: a FIXED batch -- args.config5 GiB of 64 KiB random / zero / text buffers at level
    6 -- cut over the GPUs (strong scaling).  The batch is staged in GPU 0's memory, every other rank gets
    its byte range point to point and sends its streams back (zsc_amd.sharding); the ranges are
    contiguous runs of buffers balanced by a cost proxy (sharding.cost_proxy), not by bytes.  One JSON
    line from rank 0; the scatter / gather times stand beside `value`, which times the steps only."""
    import torch
    import torch.distributed as dist
    import zsc_amd
    from zsc_amd import corpus, sharding
    from oracle.oracle_py import Oracle

    count = int(args.config5 * (1 << 30)) // 65536
    count -= count % 3
    kinds = ["random", "zero", "text"] * (count // 3)
    lens = [65536] * count
    ub, ue = sharding.scatter_assignments(sharding.cost_proxy(lens, kinds), rank, world, device=dev)
    # every rank learns every range (the root needs them to cut the payload)
    tbl = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    tbl[2 * rank], tbl[2 * rank + 1] = ub, ue
    if world > 1:
        dist.all_reduce(tbl, op=dist.ReduceOp.SUM)
    ranges = [(int(tbl[2 * r]), int(tbl[2 * r + 1])) for r in range(world)]
    mine = ue - ub
    plan = zsc_amd.DeflatePlan([65536] * mine, level=args.level)
    stride_in = plan.in_offsets[1] if mine > 1 else plan.in_bytes - 64
    stride_out = plan.out_offsets[1] if mine > 1 else plan.out_bytes - 64
    distinct = 96
    bufs = corpus.mix64k(distinct, 5)  # buffer i of the batch is bufs[i % 96]: kinds repeat with period 3
    full = None
    if rank == 0:
        host = torch.zeros(distinct * stride_in, dtype=torch.uint8)
        for i, b in enumerate(bufs):
            host[i * stride_in:i * stride_in + 65536] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        reps = (count + distinct - 1) // distinct
        full = host.to(dev).repeat(reps)[:count * stride_in]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    d_in_body = sharding.scatter_payload(full, [(b * stride_in, e * stride_in) for b, e in ranges], rank, world,
                                         device=dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    scatter_ms = (time.perf_counter() - t0) * 1e3
    d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device=dev)
    d_in[:mine * stride_in] = d_in_body
    del d_in_body, full
    d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.run(d_in.data_ptr(), d_out.data_ptr(), stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(d_in.data_ptr(), d_out.data_ptr(), stream)
    fence()
    elapsed = time.perf_counter() - t0
    olens, stats = plan.results()
    if any(x != 0 for x in stats):
        raise SystemExit(f"rank {rank}: {sum(1 for x in stats if x)} buffers failed")
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    # parity: this rank's first 96 buffers against the oracle, the rest against their first copy
    oracle = Oracle()
    ok = True
    nchk = min(distinct, mine)
    head = d_out[:nchk * stride_out].cpu()
    for k in range(nchk):
        got = bytes(head[k * stride_out:k * stride_out + olens[k]].numpy())
        ok = ok and got == oracle.compress(bufs[(ub + k) % distinct], args.level)[1]
    ok = ok and all(olens[i] == olens[i % distinct] for i in range(mine) if i % distinct < nchk)
    t0 = time.perf_counter()
    out_ranges = [(b * stride_out, e * stride_out) for b, e in ranges]
    gathered = sharding.gather_payload(d_out[:mine * stride_out], out_ranges, rank, world, device=dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    gather_ms = (time.perf_counter() - t0) * 1e3
    okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    szt = torch.tensor([sum(olens)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        dist.all_reduce(szt, op=dist.ReduceOp.SUM)
    if rank == 0:
        total_in = 65536 * count
        line = {"metric": f"deflate level-{args.level} uncompressed MB/s in, {args.config5:g} GiB of 64 KiB random / "
                          f"zero / text buffers cut over the GPUs (BASELINE config 5)",
                "value": round(total_in * args.steps / elapsed / 1e6, 2), "unit": "MB/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
                "data": "synthetic",
                "config": {"workload": f"{count} x 64 KiB buffers ({total_in} B), one third random, zero, text "
                                       f"(96 distinct, replicated), zsc_compress level {args.level}",
                           "buffers_per_gpu": [e - b for b, e in ranges],
                           "partition": "contiguous runs of buffers balanced by sharding.cost_proxy (bytes x a "
                                        "per-class cost from the measured parse rates)",
                           "parallelism": f"{world} x independent shards; payload staged on GPU 0, scattered and "
                                          "gathered point to point, outside the timed steps"},
                "staged_on_root": {"scatter_ms": round(scatter_ms, 3), "gather_ms": round(gather_ms, 3),
                                   "input_bytes": count * stride_in, "output_bytes": count * stride_out},
                "compressed_bytes_total": int(szt[0]), "all_ok": bool(int(okt[0]) == 1),
                "device": zsc_amd.device_info()}
        print(json.dumps(line), flush=True)
    plan.close()
    del gathered
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--copies", type=int, default=4096, help="Canterbury-like sets per GPU")
    ap.add_argument("--seeds", type=int, default=64,
                    help="distinct seeds among the copies (64 x 11 = 704 distinct buffers, each replicated copies / 64 times)")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=-1,
                    help="distinct buffers per rank checked against the oracle (-1: all of them)")
    ap.add_argument("--inflate-streams", type=int, default=1048576,
                    help="gzip members of the inflate section (BASELINE config 4; 0: skip)")
    ap.add_argument("--inflate-distinct", type=int, default=2048, help="distinct gzip members among them")
    ap.add_argument("--inflate-order", choices=("both", "differ", "identical"), default="both",
                    help="which decode order(s) of the replicated batch to measure (profiling aid)")
    ap.add_argument("--stage-on-root", action="store_true",
                    help="N > 1 only: the whole job's input is staged in GPU 0's memory and scattered to the "
                         "ranks point to point, the streams are gathered back (BASELINE config 5's wording; "
                         "reported beside the headline, never part of it)")
    ap.add_argument("--levels-64k", type=int, default=16384,
                    help="64 KiB buffers of the level 1/6/9 section (BASELINE config 3; 0: skip)")
    ap.add_argument("--with-inflate", action="store_true",
                    help="also time the inflate kernel on the produced streams (extra field)")
    ap.add_argument("--corpus-dir", default=os.environ.get("ZSC_CORPUS_DIR", ""),
                    help="directory holding the real Canterbury files: compress them with max_block_len 100 000 "
                         "and put the totals next to the reference's published ones (extra field)")
    ap.add_argument("--config5", type=float, default=0.0,
                    help="BASELINE config 5 instead of the headline: this many GiB (the config says 16) of 64 KiB "
                         "buffers, a fixed batch cut over the GPUs by a cost proxy, staged on GPU 0")
    ap.add_argument("--max-block-len", type=int, default=0,
                    help="also time the same buffers as streams of sections of this length "
                         "(zsc_compress with max_block_len < source_len; extra field)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import zsc_amd
    from zsc_amd import corpus, sharding

    if zsc_amd.lib.zsc_hip_init(local) != 0:
        raise SystemExit("zsc_amd: no usable gfx950 device (there is no CPU fallback)")
    if args.config5 > 0:
        run_config5(args, rank, world, local, dev)
        return

    # ---- the workload: args.copies x world Canterbury-like sets; rank 0 scatters the ranges
    seeds = max(1, min(args.seeds, args.copies))
    set_lens = [size for _, size, _ in corpus.CANTERBURY_LIKE]
    all_lens = set_lens * (args.copies * world)
    # shard at the granularity of whole sets: one unit = one 11-buffer Canterbury-like set
    unit_lens = [corpus.CANTERBURY_TOTAL] * (args.copies * world)
    ub, ue = sharding.scatter_assignments(unit_lens, rank, world, device=dev)
    begin, end = ub * len(set_lens), ue * len(set_lens)
    my_lens = all_lens[begin:end]
    my_copies = ue - ub

    plan = zsc_amd.DeflatePlan(my_lens, level=args.level)
    period_sets = [corpus.canterbury_like(s + 1000 * rank) for s in range(seeds)]
    period_bufs = [b for st in period_sets for _, b in st]
    period_bytes = plan.in_offsets[len(period_bufs)] if len(period_bufs) < len(my_lens) else plan.in_bytes - 64
    host = torch.zeros(period_bytes, dtype=torch.uint8)
    for off, b in zip(plan.in_offsets, period_bufs):
        host[off:off + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    reps = (my_copies + seeds - 1) // seeds
    d_period = host.to(dev)
    d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device=dev)
    d_in[:plan.in_bytes - 64] = d_period.repeat(reps)[:plan.in_bytes - 64]
    d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
    del d_period
    staged = None
    if args.stage_on_root and world > 1:
        # every rank owns the same number of sets, so every rank's plan has the same layout: the root
        # holds `world` images of it (here: copies of its own, the payload is what matters), scatters
        # them, and gets the output images back after the timed steps
        nb = torch.tensor([plan.in_bytes, plan.out_bytes], dtype=torch.int64, device=dev)
        lo, hi = nb.clone(), nb.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not bool((lo == hi).all()):
            raise SystemExit("--stage-on-root needs equal shards")
        in_ranges = [(r * plan.in_bytes, (r + 1) * plan.in_bytes) for r in range(world)]
        full = d_in.repeat(world) if rank == 0 else None
        fence_t0 = time.perf_counter()
        torch.cuda.synchronize()
        dist.barrier()
        d_in = sharding.scatter_payload(full, in_ranges, rank, world, device=dev)
        torch.cuda.synchronize()
        dist.barrier()
        staged = {"scatter_ms": round((time.perf_counter() - fence_t0) * 1e3, 3), "bytes_per_rank": plan.in_bytes}
        del full
    in_bytes_rank = sum(my_lens)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.run(d_in.data_ptr(), d_out.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    plan.profile(False)
    for _ in range(args.warmup):
        step()
    fence()
    plan.profile(True)
    ktimes = {k: 0.0 for k in zsc_amd.api.KERNEL_NAMES}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # per-kernel HIP-event times are read back after the timed region
    fence()
    elapsed = time.perf_counter() - t0
    lens, stats = plan.results()
    kt = plan.kernel_times_ms()  # HIP events on the launch stream, mean over the K timed steps
    for k in ktimes:
        ktimes[k] = kt[k]
    if any(s != 0 for s in stats):
        raise SystemExit(f"rank {rank}: {sum(1 for s in stats if s)} buffers failed")

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    all_sizes = sharding.gather_sizes(lens, len(all_lens), begin, rank, world, device=dev)

    # ---- parity spot check on this rank (oracle = checker only)
    from oracle.oracle_py import Oracle
    oracle = Oracle()
    out_host = None
    nver = len(period_bufs) if args.verify < 0 else min(args.verify, len(period_bufs))
    if nver:
        from concurrent.futures import ThreadPoolExecutor
        out_host = d_out[:plan.out_offsets[len(period_bufs) - 1] + plan.out_caps[len(period_bufs) - 1]].cpu()
        picks = [i if nver == len(period_bufs) else (7 * i + 2) % len(period_bufs) for i in range(nver)]

        def check(k):  # (the checker's C code runs without the interpreter lock: one oracle per call is fine)
            got = bytes(out_host[plan.out_offsets[k]:plan.out_offsets[k] + lens[k]].numpy())
            rc, want, _ = Oracle().compress(period_bufs[k], args.level)
            return k if (rc != 0 or got != want) else -1

        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
            wrong = [k for k in ex.map(check, picks) if k >= 0]
        if wrong:
            raise SystemExit(f"rank {rank}: buffer {wrong[0]} differs from the oracle")
    # the other copies: same length as, and on the device byte for byte equal to, the first copy
    nper = len(period_bufs)
    replicas_ok = all(lens[i] == lens[i % nper] for i in range(len(lens)))
    if len(my_lens) > nper:
        per_out = plan.out_offsets[nper]
        mask = torch.zeros(per_out, dtype=torch.bool)
        for k in range(nper):
            mask[plan.out_offsets[k]:plan.out_offsets[k] + lens[k]] = True
        d_mask = mask.to(dev)
        full = len(my_lens) // nper
        rows = d_out[:full * per_out].view(full, per_out)
        for r0 in range(0, full, 64):  # in slices, to bound the temporary
            replicas_ok = replicas_ok and bool((rows[r0:r0 + 64][:, d_mask] == rows[0][d_mask]).all())
        del rows, d_mask
    if not replicas_ok:
        raise SystemExit(f"rank {rank}: a replica's stream differs from its first copy")
    checked = {"distinct_buffers_vs_oracle": nver, "of": nper, "replicas_equal_on_device": bool(replicas_ok),
               "buffers": len(my_lens)}
    if staged is not None:
        out_ranges = [(r * plan.out_bytes, (r + 1) * plan.out_bytes) for r in range(world)]
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        gathered = sharding.gather_payload(d_out, out_ranges, rank, world, device=dev)
        torch.cuda.synchronize()
        dist.barrier()
        staged["gather_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
        staged["note"] = ("input images scattered from / output images gathered to GPU 0 with point-to-point "
                          "sends (zsc_amd.sharding); outside the timed steps")
        del gathered

    # ---- inflate of the streams just produced (reported beside the headline, not part of it)
    inflate_info = None
    if args.with_inflate:
        ip = zsc_amd.InflatePlan(lens, my_lens)
        d_src = torch.zeros(ip.src_bytes, dtype=torch.uint8, device=dev)
        for i in range(len(lens)):  # device-to-device staging into the inflate layout (untimed)
            d_src[ip.src_offsets[i]:ip.src_offsets[i] + lens[i]] = d_out[plan.out_offsets[i]:plan.out_offsets[i] + lens[i]]
        d_dst = torch.empty(ip.dst_bytes, dtype=torch.uint8, device=dev)
        ip.run(d_src.data_ptr(), d_dst.data_ptr(), stream); ip.results()
        fence()
        t1 = time.perf_counter()
        ip.run(d_src.data_ptr(), d_dst.data_ptr(), stream)
        fence()
        wall = time.perf_counter() - t1
        olens, used, istat, kms = ip.results()
        ok = all(s == 0 for s in istat) and olens == list(my_lens) and used == list(lens)
        chk = bytes(d_dst[ip.dst_offsets[2]:ip.dst_offsets[2] + my_lens[2]].cpu().numpy()) == period_bufs[2]
        inflate_info = {"MB_per_s_out": round(in_bytes_rank / wall / 1e6, 2), "kernel_ms": round(kms, 3),
                        "wall_ms": round(wall * 1e3, 3), "all_ok": bool(ok and chk),
                        "note": "zsc_uncompress semantics, one wave per stream, this rank only"}
        ip.close()
        del d_src, d_dst

    # ---- the same buffers as streams of sections, reported beside the headline
    sections_info = None
    if args.max_block_len > 0:
        mbl = args.max_block_len
        caps = [zsc_amd.compress_get_max_output_size2(n, mbl, args.level)[1] for n in my_lens]
        ooff, o = [], 0
        for c in caps:
            ooff.append(o)
            o += (c + 15) & ~15
        d_sec = torch.empty(o + 64, dtype=torch.uint8, device=dev)
        call = lambda: zsc_amd.compress_sections_device(d_in.data_ptr(), plan.in_offsets[:len(my_lens)], my_lens,
                                                        [mbl] * len(my_lens), d_sec.data_ptr(), ooff, caps, args.level)
        call()
        fence()
        t1 = time.perf_counter()
        rc, slens, sstat = call()
        fence()
        wall = time.perf_counter() - t1
        k = 3 % len(period_bufs)
        got = bytes(d_sec[ooff[k]:ooff[k] + slens[k]].cpu().numpy())
        want = oracle.compress(period_bufs[k], args.level, max_block_len=mbl, dest_cap=caps[k])
        sections_info = {"max_block_len": mbl, "MB_per_s_in": round(in_bytes_rank / wall / 1e6, 2),
                         "wall_ms": round(wall * 1e3, 3), "compressed_bytes": sum(slens),
                         "all_ok": bool(rc == 0 and all(x == 0 for x in sstat) and (want[0], want[1]) == (0, got)),
                         "note": "zsc_hip_compress_sections_device, streams resident in HBM, rounds + host "
                                 "simulation included, this rank only"}
        del d_sec

    if rank == 0:
        total_in = sum(all_lens)
        total_out = sum(all_sizes)
        ms_step = elapsed / args.steps * 1e3
        value = total_in * args.steps / elapsed / 1e6
        # algorithmic bytes of the parse kernel over one step on this rank: every input byte
        # read once + every stream byte written once; when the plan is cut into
        # sub-batches the kernel runs once per sub-batch and both figures are sums over them
        alg_bytes = in_bytes_rank + sum(lens)
        dom = "parse"
        dom_ms = ktimes[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        line = {
            "metric": f"deflate level-{args.level} uncompressed MB/s in, Canterbury-like x{args.copies} per GPU",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"Canterbury-like 11-buffer set (2 810 784 B, real file sizes, "
                                   f"seeded synthetic contents) x{args.copies} independent buffers per GPU "
                                   f"({seeds} distinct seeds, replicated), zsc_compress level {args.level}, "
                                   f"zlib wrapper, max_block_len >= source_len",
                       "buffers_per_gpu": len(my_lens), "input_bytes_per_gpu": in_bytes_rank,
                       "compressed_bytes_total": total_out,
                       "ratio": round(total_in / max(total_out, 1), 4),
                       "parallelism": f"{world} x independent shards, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "k_parse_seg" if args.level >= 4 else "k_parse_fast",
                         "launches_per_step": plan.sub_batches, "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "algorithmic_bytes_per_step": alg_bytes,
                         "note": "achieved = algorithmic bytes of one step / the kernel's time in that step (both "
                                 "sums over the step's launches, one per sub-batch); traffic likewise per step",
                         "kernel_ms": {k: round(v, 3) for k, v in ktimes.items()}},
            "scratch_bytes": plan.scratch_bytes,
            "checked": checked,
            "device": zsc_amd.device_info(),
        }
        line["roofline"]["limited_by"] = ("instruction issue, not HBM: the parse is serial control flow per "
                                          "position (DESIGN.md section 5)")
        # HBM traffic of the dominant kernel from rocprofv3 --pmc passes (profiles/pmc_traffic.json),
        # scaled by input bytes when the PMC run used a smaller batch of the same workload
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("level") == args.level:
                scale = in_bytes_rank / pmc["input_bytes"]
                line["roofline"]["traffic"] = round((pmc["fetch_kb"] * pmc.get("fetch_scale", 1.0) + pmc["write_kb"]) * 1024 * scale)
                line["roofline"]["traffic_note"] = pmc["note"]
        except Exception:
            pass
        if staged is not None:
            line["staged_on_root"] = staged
        if inflate_info:
            line["inflate_of_these_streams"] = inflate_info
        if sections_info:
            line["sections"] = sections_info
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.level)
    plan.close()
    del d_in, d_out
    if rank == 0 and world == 1:
        # the rest of BASELINE's metric, beside the headline (one GPU only: these are not sharded)
        torch.cuda.empty_cache()
        if args.inflate_streams > 0:
            orders = {"both": ("neighbours_differ", "neighbours_identical"), "differ": ("neighbours_differ",),
                      "identical": ("neighbours_identical",)}[args.inflate_order]
            info, members, outs = bench_inflate(dev, stream, args.inflate_streams, args.inflate_distinct, fence, orders)
            if len(orders) == 1:
                info.pop("identical_neighbours")
            try:  # HBM traffic of k_inflate from the PMC passes, scaled by output bytes
                pi = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["inflate"]
                if pi["order"] == info["order"]:
                    sc = info["output_bytes"] / pi["output_bytes"]
                    info["roofline"]["traffic"] = round((pi["fetch_kb"] * pi["fetch_scale"] + pi["write_kb"]) * 1024 * sc)
                    info["roofline"]["traffic_note"] = pi["note"]
            except Exception:
                pass
            if not args.no_cpu_baseline:
                info["cpu_baseline"] = cpu_inflate_baseline(members, outs)
            line["inflate"] = info
            torch.cuda.empty_cache()
        if args.corpus_dir:
            line["corpus"] = bench_corpus_dir(args.corpus_dir)
        if args.levels_64k > 0:
            line["levels_64k"] = bench_levels_64k(dev, stream, args.levels_64k, fence, not args.no_cpu_baseline)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
