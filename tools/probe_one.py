#!/usr/bin/env python3
"""N copies of one synthetic buffer through the device plan: python tools/probe_one.py kind size copies [level]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zsc_amd
from zsc_amd import corpus
kind, size, copies = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
level = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda", 0)
buf = corpus.make_buffer(kind, size, 1)
plan = zsc_amd.DeflatePlan([size] * copies, level=level)
stride = plan.in_offsets[1] if copies > 1 else plan.in_bytes
host = torch.zeros(stride, dtype=torch.uint8)
host[:size] = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device=dev)
d_in[:stride * copies] = host.to(dev).repeat(copies)
d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
plan.run(d_in.data_ptr(), d_out.data_ptr(), 0); plan.results()
plan.profile(True)
for _ in range(2):
    plan.run(d_in.data_ptr(), d_out.data_ptr(), 0)
lens, st = plan.results()
t = plan.kernel_times_ms()
print(f"{kind} n={size} copies={copies} L{level} out={lens[0]} parse={t['parse']:.2f} ms per-wave {size/t['parse']/1e3:.3f} MB/s total {size*copies/t['total']/1e6:.2f} GB/s", flush=True)
