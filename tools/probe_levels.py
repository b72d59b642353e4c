#!/usr/bin/env python3
"""BASELINE config 3 through the device plan: python tools/probe_levels.py [count] [levels...] (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zsc_amd
from zsc_amd import corpus
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
levels = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 6]
dev = torch.device("cuda", 0)
distinct = 96
bufs = corpus.mix64k(distinct, 5)
for level in levels:
    plan = zsc_amd.DeflatePlan([65536] * count, level=level)
    per = plan.in_offsets[distinct]
    host = torch.zeros(per, dtype=torch.uint8)
    for off, b in zip(plan.in_offsets, bufs):
        host[off:off + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    reps = (count + distinct - 1) // distinct
    d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device=dev)
    d_in[:plan.in_bytes - 64] = host.to(dev).repeat(reps)[:plan.in_bytes - 64]
    d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
    plan.run(d_in.data_ptr(), d_out.data_ptr(), 0); plan.results()
    plan.profile(True)
    for _ in range(2):
        plan.run(d_in.data_ptr(), d_out.data_ptr(), 0)
    lens, st = plan.results()
    t = plan.kernel_times_ms()
    print(f"L{level} x{count}: parse {t['parse']:.2f} ms total {t['total']:.2f} ms -> {65536*count/t['total']/1e6:.2f} GB/s (sort {t['hash_sort']:.2f} plan {t['huff_plan']:.2f}) out {sum(lens)}", flush=True)
    plan.close(); del d_in, d_out
