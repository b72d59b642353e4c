#!/usr/bin/env python3
"""Segment statistics of the segmented parser, from the host emulation (test infrastructure):
   python tools/seg_stats.py kind size [level]
Prints how the candidate batches spread over the segments of each super-step, how many
segments had to be parsed again, and what an ideal 8-wave schedule would make of it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsc_amd import corpus
E = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "tests", "emu", "libzsc_emu.so"))
kind, size = sys.argv[1], int(sys.argv[2])
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 8
data = corpus.make_buffer(kind, size, 1)
cnt = (C.c_ulonglong * 16).in_dll(E, "g_sg_cnt"); log = (C.c_uint * (1 << 20)).in_dll(E, "g_sg_log")
nlog = C.c_uint.in_dll(E, "g_sg_nlog")
E.emu_set_seg_mode(2)
E.emu_set_table(0)        # the product's defaults: no match table,
E.emu_set_stair_min(256)  # chains of 256 entries or more searched as a staircase
cap = size + (size >> 3) + 256
out = C.create_string_buffer(cap); ol = C.c_uint32()
E.emu_compress(data, size, level, 1, 0, out, cap, C.byref(ol))
recs = [(log[i], log[i + 1]) for i in range(0, nlog.value, 2)]
# split into super-steps: a queue-mode record with segment number >= previous one's starts a new step
steps = []; cur = None; prev = -1
for seg, b in recs:
    redo = seg >= 0x10000; sgn = seg & 0xffff
    if not redo and (cur is None or sgn > prev or cur["redo"]):
        cur = {"q": [], "redo": []}; steps.append(cur)
    if redo: cur["redo"].append(b)
    else: cur["q"].append(b); prev = sgn
tot_q = sum(sum(s["q"]) for s in steps); tot_r = sum(sum(s["redo"]) for s in steps)
nredo = sum(len(s["redo"]) for s in steps); nq = sum(len(s["q"]) for s in steps)
# list scheduling of the queue segments (in hand-out order) on `waves` waves; cost = batches + 8 per segment
def sched(costs):
    t = [0] * waves
    for c in costs:
        i = t.index(min(t)); t[i] += c + 8
    return max(t)
par = sum(sched(s["q"]) + sum(x + 8 for x in s["redo"]) for s in steps)
ser = tot_q + tot_r + 8 * (nq + nredo)
print(f"{kind} n={size} L{level}: steps {len(steps)} queue segs {nq} redo segs {nredo} batches queue {tot_q} redo {tot_r} "
      f"long compares {cnt[1]}  batch/byte {(tot_q+tot_r)/size:.3f}  ideal speed-up on {waves} waves {ser/par:.2f}")
# the same if the re-parsed segments had been ordinary queue segments (no serial phases)
par2 = sum(sched(s["q"] + s["redo"]) for s in steps)
print(f"  without serial re-parses: ideal speed-up {ser/par2:.2f}; serial re-parse share of the schedule {sum(sum(x + 8 for x in s['redo']) for s in steps)/par:.2f}")
