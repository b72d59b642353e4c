#!/usr/bin/env python3
"""Match-table kernel time for N copies of one synthetic buffer under several caps (GPU box):
   python tools/probe_table.py kind size copies [caps...]"""
import sys, os, subprocess
if len(sys.argv) > 4 and os.environ.get("ZSC_PROBE_CHILD") is None:
    for cap in sys.argv[4:]:
        env = dict(os.environ, ZSC_HIP_TABLE="1", ZSC_HIP_TABLE_CAP=cap, ZSC_PROBE_CHILD="1")
        subprocess.run([sys.executable, sys.argv[0]] + sys.argv[1:4], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zsc_amd
from zsc_amd import corpus
kind, size, copies = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
buf = corpus.make_buffer(kind, size, 1)
plan = zsc_amd.DeflatePlan([size] * copies, level=6)
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
gb = size * copies / 1e6
print(f"{kind} n={size} x{copies} cap={os.environ.get('ZSC_HIP_TABLE_CAP','default')}: sort {t['hash_sort']:.2f} table {t['match_table']:.2f} ms ({gb/t['match_table']:.2f} GB/s) "
      f"parse {t['parse']:.2f} ms ({gb/t['parse']:.2f} GB/s) total {t['total']:.2f} ms ({gb/t['total']:.2f} GB/s)", flush=True)
