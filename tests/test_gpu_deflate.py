"""Parity of the HIP deflate path with the oracle / golden vectors, on a real MI355X.
Everything goes through the C ABI of libzsc_hip.so (zsc_amd.api is a ctypes veneer)."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu

from zsc_amd import corpus  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
G_DEF = json.load(open(os.path.join(HERE, "golden", "deflate_golden.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def z():
    import zsc_amd
    assert zsc_amd.lib.zsc_hip_init(-1) == 0, "no usable gfx950 device: " + zsc_amd.device_info()
    print(zsc_amd.device_info())
    return zsc_amd


def test_hello_vector(z):
    # reference zsc_compress("hello hello", level 6)
    rc, out = z.compress(b"hello hello", level=6)
    assert rc == 0 and out.hex() == "789ccb48cdc9c957c800910019910449"


def test_golden_vectors_all_levels(z):
    """Streams pinned by the compiled reference (tests/golden/make_golden.py), in one batch per level."""
    for level in (1, 6, 9):
        for wb in (15, -15, 31):
            cases = [c for c in G_DEF["deflate"] if c["level"] == level and c["window_bits"] == wb]
            if not cases:
                continue
            bufs = [corpus.make_buffer(c["kind"], c["size"], c["seed"]) for c in cases]
            rc, outs, stats = z.compress_batch(bufs, level=level, window_bits=wb)
            assert rc == 0
            for c, b, o, s in zip(cases, bufs, outs, stats):
                assert sha(b) == c["in_sha256"]
                assert (s, len(o), sha(o)) == (c["rc"], c["out_len"], c["out_sha256"]), c


def test_against_oracle_edge_sizes(z, oracle):
    sizes = [0, 1, 2, 3, 4, 5, 9, 257, 258, 259, 260, 261, 262, 263, 4095, 4096, 4097, 16383, 16384,
             32505, 32506, 32507, 32767, 32768, 32769, 36863, 36864, 36865, 65273, 65274, 65275,
             65276, 65535, 65536, 65537, 65798, 98304, 131072, 200001]
    for level in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        bufs = []
        for n in sizes:
            for kind in ("text", "bitmap", "runs", "zero", "random"):
                if level not in (1, 6, 9) and kind not in ("text", "runs"):
                    continue
                bufs.append(corpus.make_buffer(kind, n, n * 5 + level))
        rc, outs, stats = z.compress_batch(bufs, level=level)
        assert rc == 0
        for b, o, s in zip(bufs, outs, stats):
            orc, want, _ = oracle.compress(b, level)
            assert s == orc == 0 and o == want, (level, len(b))


def test_strategies_and_wrappers(z, oracle):
    data = [corpus.make_buffer(k, 50000, 3) for k in ("text", "table", "object", "random")]
    for wb in (15, -15, 31):
        for strat in (0, 1, 4, 2, 3):  # default, filtered, fixed, huffman-only, rle
            rc, outs, stats = z.compress_batch(data, level=6, window_bits=wb, strategy=strat)
            assert rc == 0
            for b, o, s in zip(data, outs, stats):
                orc, want, _ = oracle.compress(b, 6, window_bits=wb, strategy=strat)
                assert s == orc == 0 and o == want, (wb, strat)


def test_one_shot_api_semantics(z, oracle):
    data = corpus.make_buffer("text", 30000, 5)
    for c in G_DEF["small"]:
        if "dest_cap" in c:
            rc, out = z.compress(data, level=6, dest_len=c["dest_cap"])
            assert (rc, len(out), sha(out)) == (c["rc"], c["out_len"], c["out_sha256"]), c
        else:
            rc, out = z.compress(data, level=6, work_len=c["work_len"])
            assert (rc, len(out)) == (c["rc"], c["out_len"]), c
    rc, out = z.compress_gzip(data, level=9)
    assert rc == 0 and out == oracle.compress(data, 9, window_bits=31)[1]
    rc, out = z.compress2(data, level=-1, window_bits=-15)
    assert rc == 0 and out == oracle.compress(data, 6, window_bits=-15)[1]


def test_canterbury_like_set_roundtrip_and_sizes(z, oracle):
    """The bench workload: 11 Canterbury-like buffers; sizes pinned by golden, bytes by the oracle,
    and the streams decode back to the input (oracle inflate = reference inflate, pinned)."""
    named = [c for c in G_DEF["deflate"] if "name" in c and c["level"] == 6]
    bufs = [corpus.make_buffer(c["kind"], c["size"], c["seed"]) for c in named]
    rc, outs, stats = z.compress_batch(bufs, level=6)
    assert rc == 0
    for c, b, o, s in zip(named, bufs, outs, stats):
        assert (s, len(o), sha(o)) == (0, c["out_len"], c["out_sha256"]), c["name"]
        assert oracle.uncompress(o, len(b)) == (0, b, len(o))


def test_device_plan_large_batch_properties(z, oracle):
    """Config-sized property test through the device-resident plan API: many buffers, results
    checked by size-independent properties (round trip through the oracle's inflate for a
    sample, identical copies give identical streams, checksum trailer = oracle adler32)."""
    import torch
    seeds = 3
    base = []
    for s in range(seeds):
        base += [b for _, b in corpus.canterbury_like(s)]
    copies = 8
    bufs = base * copies
    plan = z.DeflatePlan([len(b) for b in bufs], level=6)
    d_in = torch.zeros(plan.in_bytes, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(plan.out_bytes, dtype=torch.uint8, device="cuda")
    host = torch.zeros(plan.in_bytes, dtype=torch.uint8)
    for off, b in zip(plan.in_offsets, bufs):
        host[off:off + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    d_in.copy_(host)
    plan.profile(True)
    plan.run(d_in.data_ptr(), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    lens, stats = plan.results()
    times = plan.kernel_times_ms()
    assert times["total"] > 0 and all(s == 0 for s in stats)
    out_host = d_out.cpu()
    first = {}
    for i, (b, off, ln) in enumerate(zip(bufs, plan.out_offsets, lens)):
        stream = bytes(out_host[off:off + ln].numpy())
        k = i % len(base)
        if k in first:
            assert stream == first[k]          # same input -> same bytes, wherever it sits
        else:
            first[k] = stream
            assert stream[-4:] == oracle.adler32(b).to_bytes(4, "big")
            assert oracle.uncompress(stream, len(b)) == (0, b, ln)
            assert stream == oracle.compress(b, 6)[1]
    plan.close()


def test_config3_mix_of_64k_buffers_levels_1_6_9(z, oracle):
    """BASELINE config 3: 64 KiB random + zero + text buffers at levels 1, 6 and 9.  A batch
    the oracle can follow is compared byte for byte; a larger one is checked by round trip
    through the GPU inflate and by "same input, same stream"."""
    small = corpus.mix64k(48, seed=5)
    for level in (1, 6, 9):
        rc, outs, stats = z.compress_batch(small, level=level)
        assert rc == 0 and all(s == 0 for s in stats)
        for b, o in zip(small, outs):
            assert o == oracle.compress(b, level)[1], level
    big = corpus.mix64k(96, seed=9) * 8          # 768 buffers, 48 MiB
    for level in (1, 6, 9):
        rc, outs, stats = z.compress_batch(big, level=level)
        assert rc == 0 and all(s == 0 for s in stats)
        assert outs[:96] == outs[96:192] == outs[-96:]
        rc, back, _, st = z.uncompress_batch(outs[:96], [65536] * 96)
        assert rc == 0 and all(s == 0 for s in st) and back == big[:96]


def test_huffman_only_and_rle_golden(z):
    """Z_HUFFMAN_ONLY / Z_RLE (reference deflate_huff / deflate_rle): outcomes recorded from the
    reference, including sizes around the 16 383-symbol block cut and the window slide."""
    for st in (2, 3):
        for lvl, wb in ((6, 15), (1, 31), (9, -15)):
            cs = [c for c in G_DEF["strategies"] if (c["strategy"], c["level"], c["window_bits"]) == (st, lvl, wb)]
            bufs = [corpus.make_buffer(c["kind"], c["size"], c["seed"]) for c in cs]
            rc, outs, stats = z.compress_batch(bufs, level=lvl, window_bits=wb, strategy=st)
            assert rc == 0
            for c, o, s in zip(cs, outs, stats):
                assert (s, len(o), sha(o)) == (c["rc"], c["out_len"], c["out_sha256"]), c


def _gz_kw(spec):
    kw = dict(spec)
    if "extra" in kw:
        kw["extra"] = bytes.fromhex(kw["extra"])
    for f in ("name", "comment"):
        if f in kw:
            kw[f] = kw[f].encode()
    return kw


def test_gzip_member_header_fields(z):
    """zsc_compress_gzip / zsc_uncompress_gzip with a caller's gz_header: the
    header bytes written, the fields read back (full, truncated, small buffers), header CRC,
    tiny destinations -- all recorded from the reference."""
    data = corpus.make_buffer("text", 5000, 21)
    for c in G_DEF["gz_header"]:
        h, keep = z.gz_header_for_writing(**_gz_kw(c["spec"]))
        rc, out = z.compress2(data, level=c["level"], window_bits=31, strategy=c["strategy"], gz_header=h)
        hl = len(c["header_hex"]) // 2
        assert (rc, out[:hl].hex(), len(out), sha(out)) == (c["rc"], c["header_hex"], c["out_len"], c["out_sha256"]), c["spec"]
        for r in c["reads"]:
            src = out if r["cut"] is None else out[:r["cut"]]
            hr, bufs = z.gz_header_for_reading(*r["caps"])
            rc2, o2, used = z.uncompress2(src, len(data), 31, gz_header=hr)
            assert (rc2, len(o2), used, z.gz_header_fields(hr, bufs)) == \
                   (r["rc"], r["out_len"], r["consumed"], r["fields"]), (c["spec"], r["caps"], r["cut"])
        if "bad_last_header_byte" in c:
            bad = bytearray(out)
            bad[hl - 1] ^= 0x40
            hr, bufs = z.gz_header_for_reading(100, 100, 100)
            rc2, o2, used = z.uncompress2(bytes(bad), len(data), 31, gz_header=hr)
            b = c["bad_last_header_byte"]
            assert (rc2, len(o2), used, z.gz_header_fields(hr, bufs)) == (b["rc"], b["out_len"], b["consumed"], b["fields"])
        for sd in c.get("small_dest", []):
            h, keep = z.gz_header_for_writing(**_gz_kw(c["spec"]))
            rc3, o3 = z.compress2(data, level=c["level"], window_bits=31, gz_header=h, dest_len=sd["cap"])
            assert (rc3, o3.hex()) == (sd["rc"], sd["out_hex"]), (c["spec"], sd["cap"])
    m = G_DEF["gz_header_misc"]
    zl = z.compress2(data, level=6)[1]
    hr, bufs = z.gz_header_for_reading(10, 10, 10)
    rc2, o2, used = z.uncompress2(zl, len(data), 47, gz_header=hr)
    assert (rc2, len(o2), z.gz_header_fields(hr, bufs)) == (m["zlib_stream_auto_detect"]["rc"],
                                                          m["zlib_stream_auto_detect"]["out_len"],
                                                          m["zlib_stream_auto_detect"]["fields"])


def test_window_bits_and_mem_level(z, oracle):
    """zsc_compress2 with window_bits 9..15 (and 8 -> 9) and mem_level 1..9: golden parameter
    cases of the reference, then a sweep against the oracle (pinned on the same parameters)."""
    data = corpus.make_buffer("text", 30000, 5)
    for c in G_DEF["params"]:
        rc, out = z.compress2(data, level=c["level"], window_bits=c["window_bits"], mem_level=c["mem_level"],
                              strategy=c["strategy"], work_len=333600)  # the work size the golden calls used
        assert (rc, len(out), sha(out)) == (c["rc"], c["out_len"], c["out_sha256"]), c
    bufs = [corpus.make_buffer(k, n, n + 3) for k in ("text", "runs", "table", "bitmap", "zero")
            for n in (0, 600, 5000, 40000, 140000)]
    for wb, ml in ((9, 1), (9, 9), (10, 4), (12, 8), (14, 2), (15, 9), (15, 1), (-11, 7), (25, 3), (8, 8)):
        for level, strat in ((6, 0), (1, 0), (9, 0), (6, 3), (6, 2)):
            rc, outs, stats = z.compress_batch(bufs, level=level, window_bits=wb, mem_level=ml, strategy=strat)
            assert rc == 0
            for b, o, s in zip(bufs, outs, stats):
                orc, want, _ = oracle.compress(b, level, window_bits=wb, mem_level=ml, strategy=strat, work_len=600000)
                assert s == orc == 0 and o == want, (wb, ml, level, strat, len(b))
            if wb > 0:
                rc, back, _, st = z.uncompress_batch(outs, [len(b) for b in bufs], window_bits=wb if wb != 8 else 15)
                assert rc == 0 and all(x == 0 for x in st) and back == bufs


def test_level_0_stored(z, oracle):
    """level 0 (deflate_stored): stored blocks whose lengths follow the wrapper's output slices;
    golden cases of the reference, then sizes / section lengths / wrappers against the oracle
    (pinned on 1 900 such cases), multi-section streams with their flush markers included."""
    for c in G_DEF["stored"]:
        data = corpus.make_buffer(c["kind"], c["size"], c["seed"])
        rc, out = z.compress2(data, max_block_len=c["max_block_len"], level=0, window_bits=c["window_bits"],
                              mem_level=c["mem_level"], dest_len=c["dest_cap"], work_len=600000)
        assert (rc, len(out), sha(out)) == (c["rc"], c["out_len"], c["out_sha256"]), c
    for n in (0, 1, 507, 40000, 65531, 65536, 200000):
        data = corpus.make_buffer("object", n, n + 5)
        for wb, ml in ((15, 8), (31, 8), (-15, 8), (9, 1), (12, 9)):
            for mbl in (None, 100, 4096, 65536, 3 * n + 50):
                for cap in (None, n // 3 + 11):
                    rc, out = z.compress2(data, max_block_len=mbl, level=0, window_bits=wb, mem_level=ml,
                                          dest_len=cap, work_len=600000)
                    orc, want, _ = oracle.compress(data, 0, window_bits=wb, mem_level=ml, max_block_len=mbl,
                                                   dest_cap=cap, work_len=600000)
                    assert (rc, out) == (orc, want), (n, wb, ml, mbl, cap)
                    if rc == 0 and wb != 9:
                        assert z.uncompress2(out, n, wb)[:2] == (0, data)
    # the batch entry point stores with max_block_len = source_len
    bufs = [corpus.make_buffer("random", n, n) for n in (0, 5, 70000, 140000)]
    rc, outs, stats = z.compress_batch(bufs, level=0)
    assert rc == 0
    for b, o, s in zip(bufs, outs, stats):
        assert (s, o) == oracle.compress(b, 0)[:2]
    # with a caller's gzip header
    h, keep = z.gz_header_for_writing(name=b"stored.bin", hcrc=1, time=5)
    rc, out = z.compress2(bufs[2], level=0, window_bits=31, gz_header=h)
    assert rc == 0 and z.uncompress2(out, len(bufs[2]), 31)[:2] == (0, bufs[2])


def test_sections_levels_1_to_9(z, oracle):
    """zsc_compress with source_len > max_block_len at levels 1-9 -- sections with
    Z_FULL_FLUSH between them, the output handed out in slices of max_block_len, and the next
    section let in early where a slice runs out while a block is flushed (finding 2).  Golden
    cases of the reference through zsc_compress2, then seeded cases against the oracle (pinned
    to the reference on such cases in test_oracle.py), many streams per batched call."""
    for c in G_DEF["sections"]:
        data = corpus.make_buffer(c["kind"], c["size"], c["seed"])
        rc, out = z.compress2(data, max_block_len=c["max_block_len"], level=c.get("level", 6),
                              window_bits=c.get("window_bits", 15), strategy=c.get("strategy", 0),
                              dest_len=c.get("dest_cap"))
        assert (rc, len(out), sha(out)) == (c["rc"], c["out_len"], c["out_sha256"]), c
        if rc == 0:
            assert z.uncompress2(out, len(data), c.get("window_bits", 15))[:2] == (0, data)
    import random
    rnd = random.Random(82)
    kinds = ("text", "bitmap", "table", "random", "zero", "runs", "token", "object")
    for wb, ml, lvl, strat in ((15, 8, 6, 0), (31, 8, 9, 0), (-15, 8, 1, 0), (12, 9, 4, 1), (9, 5, 6, 0),
                               (15, 8, 3, 4), (15, 8, 6, 3), (31, 8, 2, 2)):
        bufs, mbls, caps = [], [], []
        for i in range(48):
            n = rnd.choice([300, 3000, 20000, 70000, 150000, 400000])
            mbl = rnd.choice([rnd.randrange(1, 64), rnd.randrange(64, 2000), rnd.randrange(2000, 40000),
                              rnd.randrange(20000, 100000), 65536, 32768])
            if mbl >= n:
                mbl = max(1, n // rnd.randrange(2, 6))
            if n // mbl > 300:
                mbl = n // 300 + 1
            bufs.append(corpus.make_buffer(kinds[i % len(kinds)], n, 1000 * wb + i))
            mbls.append(mbl)
            bound = oracle.max_output(n, mbl, lvl, wb, ml)[1]
            caps.append(rnd.choice([bound, bound, bound + 100, max(1, bound // 2), max(1, bound // 8)]))
        rc, outs, stats = z.compress_sections_batch(bufs, mbls, lvl, wb, ml, strat, dest_caps=caps)
        assert rc == 0, (wb, ml, lvl, strat)
        for b, m, cap, o, s in zip(bufs, mbls, caps, outs, stats):
            want = oracle.compress(b, lvl, window_bits=wb, mem_level=ml, strategy=strat, max_block_len=m,
                                   dest_cap=cap, work_len=1 << 20)
            assert (s, o) == (want[0], want[1]), (wb, ml, lvl, strat, len(b), m, cap)
    # items of one section (or none) take the same road and come out like zsc_hip_compress_batch's
    bufs = [b"", b"a", corpus.make_buffer("text", 70000, 3), corpus.make_buffer("table", 20000, 4)]
    rc, outs, stats = z.compress_sections_batch(bufs, [1, 5, 70000, 1 << 30], 6)
    assert rc == 0 and stats == [0, 0, 0, 0]
    assert outs == z.compress_batch(bufs, 6)[1]
    # a caller's gzip header in front of a stream of sections
    data = corpus.make_buffer("text", 90000, 5)
    h, keep = z.gz_header_for_writing(name=b"sections.txt", comment=b"c" * 300, hcrc=1, time=7)
    rc, out = z.compress2(data, max_block_len=7000, level=6, window_bits=31, gz_header=h)
    assert rc == 0 and z.uncompress2(out, len(data), 31)[:2] == (0, data)


def test_sections_behind_a_callers_gzip_header(z):
    """The length of a caller's gzip member header moves every output slice of a stream of
    sections (levels 0-9); outcomes recorded from the reference."""
    heads = ({"name": b"hello.txt", "time": 5},
             {"extra": bytes(range(200)) * 3, "name": b"x" * 100, "comment": b"y" * 200, "hcrc": 1})
    for c in G_DEF["gz_header_sections"]:
        data = corpus.make_buffer(c["kind"], c["size"], 5)
        h, keep = z.gz_header_for_writing(**heads[c["header"]])
        rc, out = z.compress2(data, max_block_len=c["max_block_len"], level=c["level"], window_bits=31,
                              gz_header=h, dest_len=c["dest_cap"])
        assert (rc, len(out), sha(out)) == (c["rc"], c["out_len"], c["out_sha256"]), c


def test_short_soak():
    """tools/soak.py for a quarter of a minute with a fixed seed: random sizes, classes, levels,
    wrappers, window_bits, mem_level, strategies, section lengths and dest capacities against the
    oracle, and the decoder on damaged copies of the streams (it found the hole-map and the
    data_end corners of the sections path; see DESIGN.md for the long runs)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "15", "11"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, (r.stdout[-600:], r.stderr[-600:])


@pytest.mark.gpu
def test_config_1_single_64k_buffer_one_shot(oracle):
    """BASELINE config 1, literally: ONE 65 536-byte buffer through the one-shot zsc_compress at
    level 6 (max_block_len >= source_len), compared with the oracle byte for byte, and back."""
    import zsc_amd
    for kind in ("text", "random", "zero", "table"):
        data = corpus.make_buffer(kind, 65536, 2024)
        rc, got = zsc_amd.compress(data, level=6, max_block_len=65536)
        orc, want, _ = oracle.compress(data, 6)
        assert rc == orc == 0 and got == want, kind
        rc, back, used = zsc_amd.uncompress(got, 65536)
        assert (rc, back, used) == (0, data, len(got)), kind


@pytest.mark.gpu
def test_buffer_too_long_for_32_bit_stream_positions_is_refused():
    """Bit positions inside one stream are 32-bit: a buffer whose stream could pass 2^32 bits is
    refused when the plan is made (Z_MEM_ERROR), not compressed into a corrupt stream."""
    import zsc_amd
    with pytest.raises(RuntimeError, match="-4"):
        zsc_amd.DeflatePlan([0x1ff00000], level=6)
    zsc_amd.DeflatePlan([1 << 20], level=6).close()


@pytest.mark.gpu
def test_gzip_trailer_crc_of_many_lengths(oracle):
    """The CRC-32 kernel (checksum.h) through the gzip wrapper, at lengths around its segment
    sizes (64 lanes x a power of two): the whole stream equals the oracle's."""
    import zsc_amd
    bufs = [corpus.make_buffer("random" if n % 2 else "text", n, n)
            for n in (0, 1, 15, 16, 17, 1023, 1024, 1025, 1040, 4097, 65535, 65536, 65537, 131073, 300001)]
    rc, outs, stats = zsc_amd.compress_batch(bufs, level=6, window_bits=31)
    assert rc == 0 and all(s == 0 for s in stats)
    for b, o in zip(bufs, outs):
        assert o == oracle.compress(b, 6, window_bits=31)[1], len(b)


@pytest.mark.gpu
def test_match_table_and_search_modes_on_the_device(oracle):
    """The match table (kernel 1c, opt-in: ZSC_HIP_TABLE) and the staircase search with and without its
    length threshold: the streams are the oracle's in every combination."""
    import zsc_amd
    bufs = [corpus.make_buffer(k, n, n + 1) for k, n in (("text", 200001), ("bitmap", 150000), ("table", 98304),
                                                        ("runs", 70000), ("zero", 66000), ("object", 38240),
                                                        ("random", 30000), ("text", 3100), ("text", 65275))]
    want = {}
    keep = {k: os.environ.get(k) for k in ("ZSC_HIP_TABLE", "ZSC_HIP_TABLE_CAP", "ZSC_HIP_STAIR_MIN")}
    try:
        for table, cap, stair_min in ((1, 16, 256), (1, 8, 0), (1, 64, 100000), (0, 0, 0), (0, 0, 100000), (0, 0, 256)):
            os.environ.pop("ZSC_HIP_TABLE", None)
            if table:
                os.environ["ZSC_HIP_TABLE"] = "1"
                os.environ["ZSC_HIP_TABLE_CAP"] = str(cap)
            os.environ["ZSC_HIP_STAIR_MIN"] = str(stair_min)
            for level, strat in ((6, 0), (9, 0), (4, 1), (5, 4)):
                rc, outs, stats = zsc_amd.compress_batch(bufs, level=level, strategy=strat)
                assert rc == 0
                for i, (b, o, st) in enumerate(zip(bufs, outs, stats)):
                    if (i, level, strat) not in want:
                        want[(i, level, strat)] = oracle.compress(b, level, strategy=strat)[1]
                    assert st == 0 and o == want[(i, level, strat)], (table, cap, stair_min, level, strat, i)
    finally:
        for k, v in keep.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@pytest.mark.gpu
def test_sections_many_rounds_memory_and_work_budget(oracle):
    """The corner where nearly every section end lets the next section in (incompressible data, mem_level 1
    -- blocks of 127 symbols -- and sections of a few hundred bytes): the run grows round by round.  The
    stream still equals the oracle's, the rounds whose bytes are no longer in use are given back (the
    library's log line reports the peak held), and a call that would need more parsing than its work
    budget is refused with Z_STREAM_ERROR instead of running on."""
    import subprocess, sys, textwrap
    data_args = ("random", 60000, 11)
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {repr(os.path.dirname(HERE))})
        import zsc_amd
        from zsc_amd import corpus
        from oracle.oracle_py import Oracle
        data = corpus.make_buffer{data_args}
        rc, outs, stats = zsc_amd.compress_sections_batch([data], [176], 6, 15, 1, 0)
        want = Oracle().compress(data, 6, mem_level=1, max_block_len=176, dest_cap=len(outs[0]) + 64 if outs else 1 << 20,
                                 work_len=1 << 20)
        print("RESULT", rc, stats, (outs[0] == want[1]) if rc == 0 else None)
    """)
    env = dict(os.environ, ZSC_HIP_SECTIONS_LOG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "RESULT 0 [0] True" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    peaks = [int(l.split("peak held")[1].split()[0]) for l in r.stderr.splitlines() if "peak held" in l]
    assert peaks and max(peaks) < 64 * len(corpus.make_buffer(*data_args)) + (8 << 20), peaks   # O(stream), not O(rounds x stream)
    env = dict(os.environ, ZSC_HIP_SECTIONS_BUDGET_MB="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "RESULT -2" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
