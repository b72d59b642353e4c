#!/usr/bin/env python3
"""latency of the one-shot C-ABI calls (zsc_compress / zsc_uncompress) on single buffers"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zsc_amd
from zsc_amd import corpus
for n in (1000, 65536, 1048576):
    data = corpus.make_buffer("text", n, 1)
    zsc_amd.compress(data)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        rc, comp = zsc_amd.compress(data)
    t1 = time.perf_counter()
    for _ in range(reps):
        rc2, out, used = zsc_amd.uncompress(comp, n)
    t2 = time.perf_counter()
    print(f"n={n}: zsc_compress {1e3*(t1-t0)/reps:.2f} ms/call, zsc_uncompress {1e3*(t2-t1)/reps:.2f} ms/call, ok={rc==0 and rc2==0 and out==data}", flush=True)
