#!/usr/bin/env python3
"""inflate kernel time on N copies of one synthetic buffer: python tools/probe_inflate.py kind size copies [level]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zsc_amd
from zsc_amd import corpus
kind, size, copies = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
level = int(sys.argv[4]) if len(sys.argv) > 4 else 6
buf = corpus.make_buffer(kind, size, 1)
rc, comp = zsc_amd.compress2(buf, level=level)
assert rc == 0
ip = zsc_amd.InflatePlan([len(comp)] * copies, [size] * copies)
dev = torch.device("cuda", 0)
stride = ip.src_offsets[1] if copies > 1 else ip.src_bytes
host = torch.zeros(stride, dtype=torch.uint8)
host[:len(comp)] = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
d_src = torch.zeros(ip.src_bytes, dtype=torch.uint8, device=dev)
d_src[:stride * copies] = host.to(dev).repeat(copies)
d_dst = torch.empty(ip.dst_bytes, dtype=torch.uint8, device=dev)
ip.run(d_src.data_ptr(), d_dst.data_ptr(), 0); ip.results()
ip.run(d_src.data_ptr(), d_dst.data_ptr(), 0)
olens, used, st, kms = ip.results()
ok = all(s == 0 for s in st) and bytes(d_dst[:size].cpu().numpy()) == buf
print(f"{kind} n={size} copies={copies} comp={len(comp)} inflate {kms:.2f} ms  {size*copies/kms/1e6:.2f} GB/s out  per-wave {size/kms/1e3:.3f} MB/s ok={ok}", flush=True)
