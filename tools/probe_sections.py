"""Where the time of the multi-section path goes: per call wall time on the GPU, the oracle beside it."""
import json, os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zsc_amd as z
from zsc_amd import corpus
from oracle.oracle_py import Oracle
o = Oracle()
G = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "deflate_golden.json")))
assert z.lib.zsc_hip_init(-1) == 0
for c in G["sections"]:
    data = corpus.make_buffer(c["kind"], c["size"], c["seed"])
    t0 = time.time()
    rc, out = z.compress2(data, max_block_len=c["max_block_len"], level=c.get("level", 6),
                          window_bits=c.get("window_bits", 15), strategy=c.get("strategy", 0), dest_len=c.get("dest_cap"))
    t1 = time.time()
    o.compress(data, c.get("level", 6), window_bits=c.get("window_bits", 15), strategy=c.get("strategy", 0),
               max_block_len=c["max_block_len"], dest_cap=c.get("dest_cap"))
    t2 = time.time()
    print("golden %-7s n=%6d mbl=%6d markers=%3d boundaries=%s  gpu %.3fs  oracle %.3fs" % (
        c["kind"], c["size"], c["max_block_len"], c["markers"], c.get("boundaries"), t1 - t0, t2 - t1), flush=True)
rnd = random.Random(5)
bufs = [corpus.make_buffer(("text", "table", "bitmap", "object")[i % 4], 1 << 20, i) for i in range(64)]
for mbl in (1 << 18, 1 << 16, 1 << 14):
    t0 = time.time()
    rc, outs, st = z.compress_sections_batch(bufs, [mbl] * len(bufs), 6)
    t1 = time.time()
    print("batch 64 x 1 MiB, mbl %7d: %.3fs = %.1f MB/s, rc %d" % (mbl, t1 - t0, 64 / (t1 - t0) * 1.048576, rc), flush=True)
