#!/usr/bin/env python3
"""Level 1-3 parse rate per class: python tools/probe_fast.py [level] [copies] (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zsc_amd
from zsc_amd import corpus
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda", 0)
for kind in ("random", "zero", "text", "table", "bitmap"):
    size = 65536
    buf = corpus.make_buffer(kind, size, 1)
    plan = zsc_amd.DeflatePlan([size] * copies, level=level)
    stride = plan.in_offsets[1]
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
    print(f"L{level} {kind} 64 KiB x{copies}: parse {t['parse']:.2f} ms ({size*copies/t['parse']/1e6:.2f} GB/s) total {t['total']:.2f} ms out {lens[0]}", flush=True)
    plan.close(); del d_in, d_out
