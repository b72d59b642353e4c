#!/usr/bin/env python3
"""Per-class kernel times: N copies of ONE Canterbury-like buffer per run (GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zsc_amd
from zsc_amd import corpus

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
rows = []
extra = [("zero64k", 65536, "zero"), ("random64k", 65536, "random"), ("text64k", 65536, "text")]
for i, (name, size, kind) in enumerate(list(corpus.CANTERBURY_LIKE) + extra):
    buf = corpus.make_buffer(kind, size, i)
    plan = zsc_amd.DeflatePlan([size] * copies, level=level)
    host = torch.zeros(plan.in_offsets[1] if copies > 1 else plan.in_bytes, dtype=torch.uint8)
    host[:size] = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
    d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device=dev)
    d_in[:len(host) * copies] = host.to(dev).repeat(copies)
    d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
    plan.run(d_in.data_ptr(), d_out.data_ptr(), 0); plan.results()
    plan.profile(True)
    plan.run(d_in.data_ptr(), d_out.data_ptr(), 0)
    lens, st = plan.results()
    t = plan.kernel_times_ms()
    assert all(s == 0 for s in st)
    rows.append((name, size, lens[0], t))
    print(f"{name:14s} n={size:8d} out={lens[0]:8d} table={t['match_table']:7.2f} parse={t['parse']:9.2f} ms sort={t['hash_sort']:7.2f} plan={t['huff_plan']:6.2f} emit={t['emit']:6.2f} "
          f"-> {size*copies/t['total']/1e6:8.2f} GB/s total, per-wave parse {size/t['parse']/1e3:7.3f} MB/s", flush=True)
    plan.close(); del d_in, d_out
