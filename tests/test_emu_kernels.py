"""The kernel sources (zsc_amd/csrc/*.h) executed lane by lane on the host
(tests/emu, -DZSC_WAVE_EMU) against the oracle.  Checks kernel LOGIC without a
GPU; the GPU build of the same sources is covered by the -m gpu tests."""
import ctypes as C
import os

import pytest

from zsc_amd import corpus

HERE = os.path.dirname(os.path.abspath(__file__))


# the kernel sources at two wave widths: 64 lanes, and 16 lanes -- the width group code (wave_group.h:
# the inflate decoder, optionally the segmented parser) has on the GPU, where four groups share a wave
# (group16-stage1024: the inflate decoder with a 1024-byte output ring instead of 512 bytes -- building
# it for a second size is how a read that straddled the ring's end after an inflateSync was found in
# round 2; only the inflate test runs on it)
@pytest.fixture(scope="module", params=["libzsc_emu.so", "libzsc_emu16.so", "libzsc_emu8.so", "libzsc_emu16s.so"],
                ids=["wave64", "group16", "group8", "group16-stage1024"])
def emu(request):
    L = C.CDLL(os.path.join(HERE, "emu", request.param))
    L.group16 = not request.param.endswith("emu.so")  # a narrower wave than 64 lanes
    L.group8 = request.param.endswith("8.so")
    L.inflate_only = request.param.endswith("16s.so")
    L.emu_adler32.restype = C.c_uint32
    L.emu_crc32.restype = C.c_uint32
    return L


class Rec(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("sym_begin", "sym_count", "in_begin", "in_len", "stored_ok", "last", "cut", "wend", "at")]


def emu_compress(L, data, level, wrap, strategy=0):
    n = len(data)
    cap = n + (n >> 3) + 256
    out = C.create_string_buffer(cap)
    ol = C.c_uint32()
    rc = L.emu_compress(data, n, level, wrap, strategy, out, cap, C.byref(ol))
    return rc, out.raw[:ol.value]


def parse_equals_oracle(L, oracle, data, level):
    """symbol stream and block cuts of the parse kernel against the oracle's stage P"""
    n = len(data)
    syms = (C.c_uint32 * (n + 64))()
    blocks = (Rec * (n // 16383 + 4))()
    ns, nb = C.c_uint32(), C.c_uint32()
    if L.emu_parse(data, n, level, 0, syms, C.byref(ns), blocks, C.byref(nb)) != 0:
        return False
    osy, ons, obl, onb = oracle.parse(data, level)
    if (ns.value, nb.value) != (ons, onb):
        return False
    if [syms[i] for i in range(ons)] != [(osy[i].dist << 16) | osy[i].lc for i in range(ons)]:
        return False
    return all((blocks[i].sym_begin, blocks[i].sym_count, blocks[i].in_begin, blocks[i].in_len, blocks[i].stored_ok,
                blocks[i].last) == (obl[i].sym_begin, obl[i].sym_count, obl[i].in_begin, obl[i].in_len,
                                    obl[i].stored_ok, obl[i].last) for i in range(onb))


SIZES = [0, 1, 2, 3, 4, 9, 100, 258, 259, 262, 4096, 16385, 32768, 36865, 65275, 65536, 70000]


def test_checksum_kernels(emu, oracle):
    if emu.inflate_only:
        pytest.skip("this build differs in the inflate decoder only")
    # lengths around the CRC kernel's segment sizes (64 lanes x a power of two) and Adler's 5552
    for n in (0, 1, 15, 16, 17, 1008, 1023, 1024, 1025, 1040, 2048, 2049, 4097, 5552, 65535, 65536, 65537, 70001,
              131073):
        d = corpus.make_buffer("random", n, n)
        assert emu.emu_adler32(d, n) == oracle.adler32(d)
        assert emu.emu_crc32(d, n) == oracle.crc32(d)
    d = b"\xff" * 200000  # worst case for the deferred modulo
    assert emu.emu_adler32(d, len(d)) == oracle.adler32(d)


def test_parse_kernel_symbols_and_blocks(emu, oracle):
    """hash_sort + lz_parse: the symbol stream and block cuts equal the oracle's stage P."""
    if emu.inflate_only:
        pytest.skip("this build differs in the inflate decoder only")
    if emu.group16:
        pytest.skip("the wave-per-buffer parsers are whole-wave code: 64 lanes only")
    for n in SIZES + [131072]:
        for kind in ("text", "bitmap", "table", "runs", "zero"):
            data = corpus.make_buffer(kind, n, n + 11)
            for level in (1, 3, 6, 9) if n <= 70000 else (6, 1):
                syms = (C.c_uint32 * (n + 64))()
                blocks = (Rec * (n // 16383 + 4))()
                ns, nb = C.c_uint32(), C.c_uint32()
                assert emu.emu_parse(data, n, level, 0, syms, C.byref(ns), blocks, C.byref(nb)) == 0
                osy, ons, obl, onb = oracle.parse(data, level)
                assert ns.value == ons and nb.value == onb, (n, kind, level)
                assert [syms[i] for i in range(ons)] == [(osy[i].dist << 16) | osy[i].lc for i in range(ons)]
                for i in range(onb):
                    a, b = blocks[i], obl[i]
                    assert (a.sym_begin, a.sym_count, a.in_begin, a.in_len, a.stored_ok, a.last) == \
                           (b.sym_begin, b.sym_count, b.in_begin, b.in_len, b.stored_ok, b.last)


def test_full_pipeline_streams(emu, oracle):
    """checksum + sort + parse + huffman plan + layout + emit == oracle stream, byte for byte."""
    if emu.inflate_only:
        pytest.skip("this build differs in the inflate decoder only")
    if emu.group16:
        pytest.skip("the wave-per-buffer parsers are whole-wave code: 64 lanes only")
    for n in SIZES:
        for kind in ("text", "token", "bitmap", "table", "object", "random", "zero", "runs"):
            data = corpus.make_buffer(kind, n, n + 13)
            for level, wrap, wb, strat in ((6, 1, 15, 0), (9, 1, 15, 0), (4, 0, -15, 0), (6, 2, 31, 0),
                                           (6, 1, 15, 4), (6, 1, 15, 1), (1, 1, 15, 0), (2, 2, 31, 0),
                                           (3, 0, -15, 0), (6, 1, 15, 2), (6, 2, 31, 3), (1, 1, 15, 3)):
                if n > 40000 and (level, wrap, strat) not in ((6, 1, 0), (1, 1, 0), (6, 1, 2), (6, 2, 3)):
                    continue
                rc, got = emu_compress(emu, data, level, wrap, strat)
                orc, want, _ = oracle.compress(data, level, window_bits=wb, strategy=strat)
                assert rc == orc == 0 and got == want, (n, kind, level, wrap, strat)


def test_segmented_parser_hand_over_orders(emu, oracle):
    """The multi-wave parser (lz_parse_seg.h) for long buffers: segments are parsed
    speculatively and stitched together.  Mode 2 hands segments out last-first (every
    parser finds its successors' traces, the GPU's normal case), mode 3 first-first (no
    trace is ever there in time: every hand-over is a give-up + redo).  Both must give
    the serial parse's stream, on data that resyncs at once (text), never (zero, runs),
    and on long hash chains that take the window-sweep path (bitmap)."""
    if emu.inflate_only:
        pytest.skip("this build differs in the inflate decoder only")
    if emu.group8:
        pytest.skip("the segmented parser is exercised at 64 and 16 lanes")
    try:
        for mode in (2, 3):
            emu.emu_set_seg_mode(mode)
            for n in (0, 3, 255, 256, 257, 1025, 8191, 8193, 40000, 70000, 140000):
                for kind in ("text", "bitmap", "zero", "runs", "table", "random"):
                    if n > 70000 and kind not in ("text", "bitmap", "zero"):
                        continue
                    data = corpus.make_buffer(kind, n, n + 17)
                    for level in (6, 9, 4) if n <= 40000 else (6,):
                        if emu.group16:
                            # (the Huffman and bit-packing kernels behind the parser are whole-wave code)
                            assert parse_equals_oracle(emu, oracle, data, level), (mode, n, kind, level)
                            continue
                        rc, got = emu_compress(emu, data, level, 1)
                        orc, want, _ = oracle.compress(data, level)
                        assert rc == orc == 0 and got == want, (mode, n, kind, level)
    finally:
        emu.emu_set_seg_mode(0)


def test_match_table_entries_against_the_reference_walk(emu, oracle):
    """match_table.h: every entry the table kernel completes equals longest_match done the
    reference's way (the walk along the position's own chain with its chain budget), for
    prev_length 2 and -- where the entry says it answers for longer ones too -- for the length the
    lazy parse would ask with.  What the kernel leaves incomplete the parser searches itself."""
    if emu.inflate_only or emu.group16:
        pytest.skip("the table kernel is whole-wave code: 64 lanes only")
    inc = C.c_uint32()
    try:
        for cap in (8, 16, 64):
            emu.emu_set_table_cap(cap)
            for kind, n, level, strat in (("text", 70001, 6, 0), ("bitmap", 40000, 6, 0), ("runs", 30000, 9, 0),
                                          ("table", 66000, 4, 0), ("object", 38240, 6, 1), ("zero", 70000, 6, 0),
                                          ("random", 5000, 8, 0), ("text", 300, 5, 4), ("token", 3, 6, 0),
                                          ("text", 0, 6, 0)):
                if cap != 16 and n > 40000:
                    continue
                data = corpus.make_buffer(kind, n, n + cap)
                bad = emu.emu_table_check(data, n, level, strat, C.byref(inc))
                assert bad == 0, (cap, kind, n, level, strat)
                assert n < 1000 or inc.value < n   # (and something is complete)
    finally:
        emu.emu_set_table_cap(16)


def test_segmented_parser_search_modes(emu, oracle):
    """The segmented parser's ways to a match give the reference's stream: hops over the match
    table (on / off), the staircase search over the shortest chains (for every chain / for long ones
    only, the product's default / never), chains below the chain budget searched by all lanes at once
    (the product's default) or walked like the others, the reference's own walk; chain lengths and the links
    into the previous tile worked out by the parser from the bucket directories or read from k_link_prev's arrays."""
    if emu.inflate_only or emu.group8:
        pytest.skip("the segmented parser is exercised at 64 and 16 lanes")
    try:
        emu.emu_set_seg_mode(2)
        for table, stair_min, one, link in ((1, 0, 1, 1), (1, 256, 1, 1), (1, 0xffffffff, 1, 1), (0, 0, 1, 1), (0, 256, 1, 1),
                                            (0, 0xffffffff, 1, 1), (0, 256, 0, 1), (0, 0xffffffff, 0, 0), (1, 256, 0, 0),
                                            (0, 256, 1, 0)):
            emu.emu_set_table(table)
            emu.emu_set_stair_min(stair_min)
            emu.emu_set_one(one)
            emu.emu_set_link_in_parser(link)  # 0: chain lengths and links from k_link_prev's arrays
            for kind, n in (("text", 90000), ("bitmap", 70000), ("table", 40000), ("runs", 30000), ("zero", 70000),
                            ("random", 20000), ("object", 38240), ("text", 65275)):
                data = corpus.make_buffer(kind, n, n + table)
                for level in (6, 9, 4) if n <= 40000 else (6,):
                    if emu.group16:
                        assert parse_equals_oracle(emu, oracle, data, level), (table, stair_min, one, link, kind, n, level)
                        continue
                    rc, got = emu_compress(emu, data, level, 1)
                    orc, want, _ = oracle.compress(data, level)
                    assert rc == orc == 0 and got == want, (table, stair_min, one, link, kind, n, level)
    finally:
        emu.emu_set_seg_mode(0)
        emu.emu_set_table(1)
        emu.emu_set_stair_min(0)
        emu.emu_set_one(1)
        emu.emu_set_link_in_parser(1)


def test_greedy_parser_with_and_without_an_lds_window(emu, oracle):
    """Levels 1-3 (deflate_fast): the parser that reads the window from the input itself (the product's
    choice) and the one with an LDS ring give the oracle's streams, through window slides and at the
    edges of the input."""
    if emu.inflate_only or emu.group16:
        pytest.skip("the wave-per-buffer parsers are whole-wave code: 64 lanes only")
    try:
        for glob in (1, 0):
            emu.emu_set_fast_global(glob)
            for n in (0, 1, 3, 4, 262, 4097, 40959, 40960, 40961, 65275, 65536, 70001, 131072):
                for kind in ("text", "runs", "random", "zero", "bitmap"):
                    data = corpus.make_buffer(kind, n, n + 19)
                    for level in (1, 2, 3):
                        if n > 70001 and level != 1:
                            continue
                        rc, got = emu_compress(emu, data, level, 1)
                        orc, want, _ = oracle.compress(data, level)
                        assert rc == orc == 0 and got == want, (glob, n, kind, level)
    finally:
        emu.emu_set_fast_global(1)


def emu_uncompress(L, data, cap, wb):
    dst = C.create_string_buffer(max(cap, 1))
    ol, used = C.c_uint32(), C.c_uint32()
    rc = L.emu_uncompress(data, len(data), wb, dst, cap, C.byref(ol), C.byref(used))
    return rc, dst.raw[:ol.value], used.value


def test_inflate_kernel_including_resynchronisation(emu, oracle):
    """inflate.h lane by lane: the reference's known answers, round trips, and damaged
    multi-section streams (inflateSync recovery) from the golden fixtures, against the
    recorded outcome of the reference and against the oracle."""
    import hashlib
    import json
    from test_oracle import apply_edits
    g = json.load(open(os.path.join(HERE, "golden", "inflate_golden.json")))
    for c in g["inflate_kat"]:
        raw = bytes(int(x, 16) for x in c["hex"].split())
        rc, out, used = emu_uncompress(emu, raw, c["dest_cap"], c["window_bits"])
        assert (rc, out.hex(), used) == (c["rc"], c["out_hex"], c["consumed"]), c
    for r in g["resync"]:
        stream = bytes.fromhex(r["stream_hex"])
        data = corpus.make_buffer(r["kind"], r["size"], r["seed"])
        assert emu_uncompress(emu, stream, len(data), r["window_bits"]) == (0, data, len(stream))
        for c in r["cases"]:
            rc, out, used = emu_uncompress(emu, apply_edits(stream, c["edits"]), c["dest_cap"], r["window_bits"])
            assert (rc, len(out), used, hashlib.sha256(out).hexdigest()) == \
                   (c["rc"], c["out_len"], c["consumed"], c["out_sha256"]), c
    # seeded damage to single-section streams of every wrapper: kernel == oracle
    import random
    rnd = random.Random(9)
    for i in range(300):
        n = rnd.choice([200, 3000, 20000])
        data = corpus.make_buffer(("text", "zero", "table", "random", "runs")[i % 5], n, i)
        wb = (15, 31, -15)[i % 3]
        comp = bytearray(oracle.compress(data, (1, 6, 9)[i % 3], window_bits=wb)[1])
        for _ in range(rnd.randrange(1, 3)):
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        if i % 7 == 0:
            del comp[rnd.randrange(len(comp)):]
        cap = rnd.choice([n, n + 50, max(1, n // 2)])
        assert emu_uncompress(emu, bytes(comp), cap, wb) == oracle.uncompress(bytes(comp), cap, wb), (i, wb, cap)


def test_window_bits_and_mem_level(emu, oracle):
    """zsc_compress2's window_bits 9..15 and mem_level 1..9: a smaller window
    moves MAX_DIST and the slide points, mem_level the block cut (lit_bufsize) and -- in one
    corner -- which candidate at exactly MAX_DIST heads its chain (LZ_HEAD_BLOCKED)."""
    if emu.inflate_only:
        pytest.skip("this build differs in the inflate decoder only")
    if emu.group16:
        pytest.skip("the wave-per-buffer parsers are whole-wave code: 64 lanes only")
    try:
        for wb, ml in ((9, 1), (9, 9), (10, 4), (12, 8), (14, 2), (15, 9), (15, 1), (11, 7)):
            emu.emu_set_params(wb, ml)
            for n in (0, 3, 513, 763, 5000, 20000, 70000):
                for kind in ("text", "runs", "table", "bitmap"):
                    data = corpus.make_buffer(kind, n, n * 3 + wb)
                    for level, strat in ((6, 0), (1, 0), (9, 0), (6, 3), (6, 2)):
                        if n > 20000 and level == 9:
                            continue
                        for mode in (0, 2) if level >= 4 and strat == 0 else (0,):
                            emu.emu_set_seg_mode(mode)
                            rc, got = emu_compress(emu, data, level, 1, strat)
                            orc, want, _ = oracle.compress(data, level, window_bits=wb, mem_level=ml,
                                                           strategy=strat, work_len=600000)
                            assert rc == orc == 0 and got == want, (wb, ml, n, kind, level, strat, mode)
    finally:
        emu.emu_set_params(15, 8)
        emu.emu_set_seg_mode(0)


def emu_compress_sections(L, data, mbl, level, wb, strategy, cap, mem_level=8):
    wrap, w = (0, -wb) if wb < 0 else (2, wb - 16) if wb > 15 else (1, wb)
    L.emu_set_params(w, mem_level)
    out = C.create_string_buffer(cap + 64)
    ol, parses = C.c_uint32(), C.c_uint32()
    rc = L.emu_compress_sections(data, len(data), mbl, level, wrap, strategy, out, cap, C.byref(ol),
                                 C.byref(parses))
    L.emu_set_params(15, 8)
    return rc, out.raw[:ol.value], parses.value


def test_sections_rounds_and_joints(emu, oracle):
    """zsc_compress with source_len > max_block_len.  Round 0 parses every section
    on its own; the host follows the wrapper's output slices (sections.h) and has a run parsed
    again wherever a slice ran out at a place that lets the next section in early (finding 2).
    The result equals the oracle's call-by-call restatement (pinned to the reference in
    test_oracle.py), also where dest is too small and only a prefix is handed back."""
    if emu.inflate_only:
        pytest.skip("this build differs in the inflate decoder only")
    if emu.group16:
        pytest.skip("the wave-per-buffer parsers are whole-wave code: 64 lanes only")
    import random
    rnd = random.Random(81)
    kinds = ("text", "bitmap", "table", "random", "zero", "runs", "token", "object")
    rejoined = 0
    for it in range(70):
        n = rnd.choice([300, 3000, 20000, 70000, 150000])
        data = corpus.make_buffer(rnd.choice(kinds), n, it)
        mbl = rnd.choice([rnd.randrange(1, 64), rnd.randrange(64, 2000), rnd.randrange(2000, 40000),
                          rnd.randrange(20000, 100000)])
        if mbl >= n:
            mbl = max(1, n // rnd.randrange(2, 6))
        if n // mbl > 300:
            mbl = n // 300 + 1
        lvl = rnd.choice([1, 3, 4, 6, 9])
        wb = rnd.choice([15, 15, 31, -15, 12, 9])
        ml = rnd.choice([8, 8, 8, 9, 5, 1, 2])
        strat = rnd.choice([0, 0, 1, 4, 2, 3]) if ml >= 8 else 0   # the reference asserts on Z_FIXED + small mem_level
        bound = oracle.max_output(n, mbl, lvl, wb, ml)[1]
        cap = rnd.choice([bound, bound, bound + 100, max(1, bound // 2), max(1, bound // 8)])
        want = oracle.compress(data, lvl, window_bits=wb, mem_level=ml, strategy=strat, max_block_len=mbl,
                               dest_cap=cap, work_len=1 << 20)
        rc, out, parses = emu_compress_sections(emu, data, mbl, lvl, wb, strat, cap, ml)
        assert (rc, out) == (want[0], want[1]), (it, n, mbl, lvl, wb, ml, strat, cap)
        rejoined += parses - (n + mbl - 1) // mbl
    assert rejoined > 150   # runs parsed again: joints are the rule, not the exception (most are speculated)
    # found by tools/soak.py on the GPU: with mem_level 1 a block fills up (127 symbols) exactly where a
    # 164-byte section ends; the next section is let in while that block is flushed, so nobody catches up
    # on the section's last two positions (no s->insert): they never enter the reference's hash chains
    data = corpus.make_buffer("object", 65535, 978947581)[:26000]
    bound = oracle.max_output(len(data), 164, 6, 15, 1)[1]
    want = oracle.compress(data, 6, mem_level=1, max_block_len=164, dest_cap=bound, work_len=1 << 20)
    assert emu_compress_sections(emu, data, 164, 6, 15, 0, bound, 1)[:2] == (want[0], want[1])
    for seed in (3, 4):
        for ml in (1, 2):
            data = corpus.make_buffer("random", 40000, seed)   # runs that outgrow the small rings
            bound = oracle.max_output(len(data), 376, 4, -15, ml)[1]
            want = oracle.compress(data, 4, window_bits=-15, mem_level=ml, max_block_len=376, dest_cap=bound,
                                   work_len=1 << 20)
            assert emu_compress_sections(emu, data, 376, 4, -15, 0, bound, ml)[:2] == (want[0], want[1])
