#!/usr/bin/env python3
"""One stream through the sections path with the library's log (rounds, bytes parsed, device memory held):
   ZSC_HIP_SECTIONS_LOG=1 python tools/probe_sections_one.py kind n max_block_len [level wbits mem_level strategy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsc_amd import corpus
import zsc_amd as z
kind, n, mbl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lvl, wb, ml, strat = (int(x) for x in (sys.argv[4:8] + ["6", "15", "8", "0"][len(sys.argv) - 4:]))
data = corpus.make_buffer(kind, n, 12345)
assert z.lib.zsc_hip_init(-1) == 0
for rep in range(2):
    t = time.time()
    rc, outs, stats = z.compress_sections_batch([data], [mbl], lvl, wb, ml, strat)
    print(f"{kind} n={n} max_block_len={mbl} level {lvl} wbits {wb} mem_level {ml} strategy {strat}: rc {rc} status {stats[0]} out {len(outs[0])} in {time.time() - t:.2f} s", flush=True)
