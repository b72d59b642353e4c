#!/usr/bin/env python3
"""Match-table statistics from the host emulation (test infrastructure):
   python tools/table_stats.py kind size [level]
How many table entries are left incomplete, how many loop tops of the segmented parser were hops
and how many went through the search, and whether the stream equals the oracle's."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsc_amd import corpus
from oracle.oracle_py import Oracle
E = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "tests", "emu", "libzsc_emu.so"))
kind, size = sys.argv[1], int(sys.argv[2])
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
use_table = int(sys.argv[4]) if len(sys.argv) > 4 else 1
data = corpus.make_buffer(kind, size, 1)
sg = (C.c_ulonglong * 16).in_dll(E, "g_sg_cnt"); mt = (C.c_ulonglong * 4).in_dll(E, "g_mt_cnt")
E.emu_set_seg_mode(2)
E.emu_set_table(use_table)
cap = size + (size >> 3) + 256
out = C.create_string_buffer(cap); ol = C.c_uint32()
rc = E.emu_compress(data, size, level, 1, 0, out, cap, C.byref(ol))
orc, want, _ = Oracle().compress(data, level)
print(f"{kind} n={size} L{level}: stream {'==' if out.raw[:ol.value] == want else '!='} oracle ({ol.value} B)")
print(f"  table: incomplete {mt[1]/max(1,mt[0]):.4f} of the entries; answering for longer prev_lengths too {mt[2]/max(1,mt[0]):.4f}")
print(f"  parser: loop tops {sg[4]} ({sg[4]/size:.3f}/byte), hops {sg[7]} ({sg[7]/max(1,sg[4]):.3f} of them), searches {sg[5]}, batches {sg[0]} ({sg[0]/size:.3f}/byte), long compares {sg[1]}, walks done the reference's way {sg[8]} ({sg[11]} of the batches), lazy searches skipped (an empty chain) {sg[9]}, single look-ups in the table {sg[10]}")
