#!/usr/bin/env python3
"""Generate tests/golden/*.json from the compiled reference (oracle/_ref/libzsc_ref.so).

Run in the build container (needs /root/reference to compile the reference):
    python tests/golden/make_golden.py
The fixtures hold inputs as (kind, size, seed) triples of zsc_amd.corpus (plus the
input's SHA-256, so generator drift is detected) and the reference's outputs as
length + SHA-256, with a few complete small streams in hex.  Inflate known-answer
inputs are the hex strings the reference's own test/infcover.c feeds to inflate
(:367-371, :399-411, :583-613, :643-658); the expected results come from running
the reference's zsc_uncompress2 on them here.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle_py import Reference, build  # noqa: E402
from zsc_amd import corpus  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def apply_edits(stream: bytes, ops) -> bytes:
    """["f", pos, mask] xor a byte, ["x", pos, hex] overwrite, ["d", pos, count] delete, ["t", len] truncate"""
    b = bytearray(stream)
    for op in ops:
        if op[0] == "f":
            b[op[1]] ^= op[2]
        elif op[0] == "x":
            raw = bytes.fromhex(op[2])
            b[op[1]:op[1] + len(raw)] = raw
        elif op[0] == "d":
            del b[op[1]:op[1] + op[2]]
        else:
            del b[op[1]:]
    return bytes(b)


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


KINDS = ["text", "token", "bitmap", "table", "object", "random", "zero", "runs"]
SIZES = [0, 1, 2, 3, 258, 259, 4096, 16383, 32768, 40000, 65274, 65275, 65535, 65536, 65537,
         98304, 150000]

# hex inputs of reference test/infcover.c: (hex, window_bits, dest_cap)
INFCOVER = [
    ("63 0", -15, 1), ("63 18 5", -8, 259), ("63 18 68 30 d0 0 0", -8, 259), ("3 0", -15, 1),
    ("1f 8b 0 0", 31, 0), ("1f 8b 8 80", 31, 0), ("77 85", 15, 0), ("8 99", 0, 0), ("78 9c", 8, 0),
    ("78 9c 63 0 0 0 1 0 1", 15, 1),
    ("1f 8b 8 1e 0 0 0 0 0 0 1 0 0 0 0 0 0", 47, 1),
    ("1f 8b 8 2 0 0 0 0 0 0 1d 26 3 0 0 0 0 0 0 0 0 0", 47, 0),
    ("78 90", 47, 0), ("8 b8 0 0 0 1", 8, 0), ("78 9c 63 0", 15, 1),
    ("0 0 0 0 0", -15, 300), ("3 0", -15, 300), ("6", -15, 300), ("1 1 0 fe ff 0", -15, 300),
    ("fc 0 0", -15, 300), ("4 0 fe ff", -15, 300), ("4 0 24 49 0", -15, 300),
    ("4 0 24 e9 ff ff", -15, 300), ("4 0 24 e9 ff 6d", -15, 300),
    ("4 80 49 92 24 49 92 24 71 ff ff 93 11 0", -15, 300),
    ("4 80 49 92 24 49 92 24 f b4 ff ff c3 84", -15, 300),
    ("4 c0 81 8 0 0 0 0 20 7f eb b 0 0", -15, 300), ("2 7e ff ff", -15, 300),
    ("c c0 81 0 0 0 0 0 90 ff 6b 4 0", -15, 300),
    ("1f 8b 8 0 0 0 0 0 0 0 3 0 0 0 0 1", 47, 300),
    ("1f 8b 8 0 0 0 0 0 0 0 3 0 0 0 0 0 0 0 0 1", 47, 300),
    ("5 c0 21 d 0 0 0 80 b0 fe 6d 2f 91 6c", -15, 300),
    ("5 e0 81 91 24 cb b2 2c 49 e2 f 2e 8b 9a 47 56 9f fb fe ec d2 ff 1f", -15, 300),
    ("ed c0 1 1 0 0 0 40 20 ff 57 1b 42 2c 4f", -15, 600),
    ("ed cf c1 b1 2c 47 10 c4 30 fa 6f 35 1d 1 82 59 3d fb be 2e 2a fc f c", -15, 600),
    ("ed c0 81 0 0 0 0 80 a0 fd a9 17 a9 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 "
     "0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 6", -15, 600),
    ("2 8 20 80 0 3 0", -15, 258), ("63 18 5 40 c 0", -8, 300),
    ("e5 e0 81 ad 6d cb b2 2c c9 01 1e 59 63 ae 7d ee fb 4d fd b5 35 41 68 ff 7f 0f 0 0 0", -8, 258),
    ("25 fd 81 b5 6d 59 b6 6a 49 ea af 35 6 34 eb 8c b9 f6 b9 1e ef 67 49 50 fe ff ff 3f 0 0", -8, 258),
    ("3 7e 0 0 0 0 0", -8, 258), ("1b 7 0 0 0 0 0", -8, 258),
    ("d c7 1 ae eb 38 c 4 41 a0 87 72 de df fb 1f b8 36 b1 38 5d ff ff 0", -8, 258),
    ("63 18 5 8c 10 8 0 0 0 0", -8, 259),
    ("63 60 60 18 c9 0 8 18 18 18 26 c0 28 0 29 0 0 0", -8, 259),
    ("63 0 3 0 0 0 0 0", -8, 259),
]


def main():
    build(ref=True)
    R = Reference()
    deflate_cases = []
    streams = []
    for size in SIZES:
        for kind in KINDS:
            seed = size * 3 + 1
            data = corpus.make_buffer(kind, size, seed)
            for level in (1, 6, 9) if size <= 70000 else (6,):
                for wb in (15, -15, 31):
                    if wb != 15 and (level != 6 or kind not in ("text", "random")):
                        continue
                    rc, out = R.compress(data, level, window_bits=wb)
                    case = {"kind": kind, "size": size, "seed": seed, "level": level,
                            "window_bits": wb, "in_sha256": sha(data), "rc": rc,
                            "out_len": len(out), "out_sha256": sha(out)}
                    deflate_cases.append(case)
                    if size in (0, 1, 3, 258, 259) and kind in ("text", "zero") and wb == 15:
                        streams.append({**case, "in_hex": data.hex(), "out_hex": out.hex()})
    # the Canterbury-like set (bench workload), level 6 and 9 sizes for seed 0
    for i, (name, size, kind) in enumerate(corpus.CANTERBURY_LIKE):
        data = corpus.make_buffer(kind, size, i)
        for level in (1, 6, 9):
            rc, out = R.compress(data, level)
            deflate_cases.append({"kind": kind, "size": size, "seed": i, "level": level,
                                  "window_bits": 15, "in_sha256": sha(data), "rc": rc,
                                  "out_len": len(out), "out_sha256": sha(out), "name": name})
    # other strategies / error codes through the same entry point
    data = corpus.make_buffer("text", 30000, 5)
    params = []
    for wb, ml, st, lvl in [(15, 8, 1, 6), (15, 8, 4, 6), (15, 8, 1, 9), (12, 8, 0, 6), (15, 9, 0, 6),
                            (15, 1, 0, 6), (-9, 4, 0, 6), (25, 8, 0, 6), (8, 8, 0, 6), (-8, 8, 0, 6),
                            (15, 0, 0, 6), (15, 10, 0, 6), (7, 8, 0, 6), (15, 8, 5, 6), (15, 8, 0, 10),
                            (15, 8, 0, -1)]:
        rc, out = R.compress(data, lvl, window_bits=wb, mem_level=ml, strategy=st)
        params.append({"window_bits": wb, "mem_level": ml, "strategy": st, "level": lvl, "rc": rc,
                       "out_len": len(out), "out_sha256": sha(out)})
    # Z_HUFFMAN_ONLY (2) and Z_RLE (3): their own parse functions, src/deflate.c:2129-2247
    strategies = []
    for kind, size, seed in (("text", 30000, 5), ("runs", 70000, 6), ("zero", 66000, 7), ("bitmap", 140000, 8),
                             ("random", 16384, 9), ("table", 3, 10), ("text", 0, 11)):
        d2 = corpus.make_buffer(kind, size, seed)
        for st in (2, 3):
            for lvl, wb in ((6, 15), (1, 31), (9, -15)):
                rc, out = R.compress(d2, lvl, window_bits=wb, strategy=st)
                strategies.append({"kind": kind, "size": size, "seed": seed, "strategy": st, "level": lvl,
                                   "window_bits": wb, "rc": rc, "out_len": len(out), "out_sha256": sha(out)})
    # caller-supplied gzip member headers (write: src/deflate.c:1091-1200, read: src/inflate.c:786-954)
    from zsc_amd.api import gz_header_for_writing, gz_header_for_reading, gz_header_fields
    gzh = []
    d3 = corpus.make_buffer("text", 5000, 21)
    plain = R.compress(d3, 6, window_bits=31)[1]
    for k, spec in enumerate((
            {"text": 1, "time": 0x12345678, "os": 7, "name": "hello.txt", "hcrc": 0},
            {"text": 0, "time": 1, "os": 255, "comment": "a comment", "hcrc": 1},
            {"text": 1, "time": 0xffffffff, "os": 3, "extra": "00112233445566", "name": "n", "comment": "", "hcrc": 1},
            {"text": 0, "time": 0, "os": 0, "extra": "", "hcrc": 0},
            {"text": 0, "time": 77, "os": 3, "extra": "ab" * 300, "name": "x" * 100, "comment": "y" * 200, "hcrc": 1},
            {"text": 0, "time": 0, "os": 3, "hcrc": 0})):
        kw = dict(spec)
        for f in ("extra",):
            if f in kw:
                kw[f] = bytes.fromhex(kw[f])
        for f in ("name", "comment"):
            if f in kw:
                kw[f] = kw[f].encode()
        for lvl, strat in ((6, 0), (9, 0), (1, 0), (6, 3), (0, 0)):
            h, keep = gz_header_for_writing(**kw)
            rc, out = R.compress(d3, lvl, window_bits=31, strategy=strat, gz_header=h)
            body = R.compress(d3, lvl, window_bits=31, strategy=strat)[1]
            hlen = 10 + (2 + len(kw["extra"]) if "extra" in kw else 0) + (len(kw["name"]) + 1 if "name" in kw else 0) \
                + (len(kw["comment"]) + 1 if "comment" in kw else 0) + (2 if kw.get("hcrc") else 0)
            assert rc == 0 and (lvl == 0 or out[hlen:] == body[10:])
            case = {"spec": spec, "level": lvl, "strategy": strat, "rc": rc, "header_hex": out[:hlen].hex(),
                    "out_len": len(out), "out_sha256": sha(out), "reads": []}
            if lvl == 6 and strat == 0:
                for caps in ((0, 0, 0), (4, 5, 3), (1000, 1000, 1000)):
                    for cut in (None, 3, 11, hlen - 1, hlen + 5):
                        src = out if cut is None else out[:cut]
                        hr, bufs = gz_header_for_reading(*caps)
                        rc2, o2, used = R.uncompress(src, len(d3), 31, gz_header=hr)
                        case["reads"].append({"caps": caps, "cut": cut, "rc": rc2, "out_len": len(o2),
                                              "consumed": used, "fields": gz_header_fields(hr, bufs)})
                # a damaged header crc, a zlib stream read with a gz_header, a small dest
                bad = bytearray(out)
                bad[hlen - 1] ^= 0x40
                hr, bufs = gz_header_for_reading(100, 100, 100)
                rc2, o2, used = R.uncompress(bytes(bad), len(d3), 31, gz_header=hr)
                case["bad_last_header_byte"] = {"rc": rc2, "out_len": len(o2), "consumed": used,
                                                "fields": gz_header_fields(hr, bufs)}
                for cap in (0, 5, hlen - 1, hlen, hlen + 7):
                    h, keep = gz_header_for_writing(**kw)
                    rc3, o3 = R.compress(d3, lvl, window_bits=31, gz_header=h, dest_cap=cap)
                    case.setdefault("small_dest", []).append({"cap": cap, "rc": rc3, "out_hex": o3.hex()})
            gzh.append(case)
    # a caller's header in front of a stream of sections: its length moves every output slice
    gz_sections = []
    for k, kw in enumerate(({"name": b"hello.txt", "time": 5},
                            {"extra": bytes(range(200)) * 3, "name": b"x" * 100, "comment": b"y" * 200, "hcrc": 1})):
        hlen = 10 + (2 + len(kw["extra"]) if "extra" in kw else 0) + (len(kw["name"]) + 1 if "name" in kw else 0) \
            + (len(kw["comment"]) + 1 if "comment" in kw else 0) + (2 if kw.get("hcrc") else 0)
        for kind, size in (("text", 90000), ("random", 20000)):
            d5 = corpus.make_buffer(kind, size, 5)
            for mbl, lvl, cap in ((7000, 6, None), (300, 1, None), (32768, 9, None), (7000, 0, None),
                                  (7000, 6, hlen + 5000), (300, 6, hlen - 3)):
                h, keep = gz_header_for_writing(**kw)
                if cap is None:  # the slices depend on how much dest is left: say it
                    cap = R.max_output(size, mbl, lvl, 31)[1] + hlen + 100
                rc, out = R.compress(d5, lvl, window_bits=31, max_block_len=mbl, gz_header=h, dest_cap=cap)
                gz_sections.append({"header": k, "kind": kind, "size": size, "max_block_len": mbl, "level": lvl,
                                    "dest_cap": cap, "rc": rc,
                                    "out_len": len(out), "out_sha256": sha(out)})
    hr, bufs = gz_header_for_reading(10, 10, 10)
    zl = R.compress(d3, 6, window_bits=15)[1]
    rc2, o2, used = R.uncompress(zl, len(d3), 47, gz_header=hr)
    gz_misc = {"zlib_stream_auto_detect": {"rc": rc2, "out_len": len(o2), "fields": gz_header_fields(hr, bufs)},
               "header_on_zlib_compress": R.compress(d3, 6, window_bits=15, gz_header=gz_header_for_writing()[0])[0],
               "header_on_zlib_uncompress": R.uncompress(zl, len(d3), 15, gz_header=gz_header_for_reading()[0])[0]}
    # level 0 (deflate_stored): block lengths follow the wrapper's output slices
    stored = []
    for kind, size, seed in (("random", 0, 1), ("random", 1, 2), ("text", 507, 3), ("text", 32768, 4),
                             ("random", 65531, 5), ("random", 65536, 6), ("table", 70000, 7), ("bitmap", 300000, 8)):
        d5 = corpus.make_buffer(kind, size, seed)
        for wb, ml, mbl, cap in ((15, 8, None, None), (31, 8, None, None), (-15, 8, None, None), (15, 1, None, None),
                                 (12, 9, None, None), (15, 8, None, size // 2 + 7), (15, 8, 2 * size + 100, None)):
            rc, out = R.compress(d5, 0, window_bits=wb, mem_level=ml, max_block_len=mbl, dest_cap=cap, work_len=600000)
            stored.append({"kind": kind, "size": size, "seed": seed, "window_bits": wb, "mem_level": ml,
                           "max_block_len": mbl, "dest_cap": cap, "rc": rc, "out_len": len(out), "out_sha256": sha(out)})
    small = []
    for cap in (0, 1, 2, 100, 12000, 40000):
        rc, out = R.compress(data, 6, dest_cap=cap)
        small.append({"dest_cap": cap, "rc": rc, "out_len": len(out), "out_sha256": sha(out)})
    for wl in (0, 333599, 333600):
        rc, out = R.compress(data, 6, work_len=wl)
        small.append({"work_len": wl, "rc": rc, "out_len": len(out)})
    # multi-section behaviour (SURVEY finding 2): max_block_len < source_len
    sections = []
    big = corpus.make_buffer("text", 300000, 9)
    for mbl in (20000, 100000, 299999):
        rc, out = R.compress(big, 6, max_block_len=mbl)
        sections.append({"kind": "text", "size": 300000, "seed": 9, "max_block_len": mbl, "rc": rc,
                         "out_len": len(out), "out_sha256": sha(out),
                         "markers": out.count(b"\x00\x00\xff\xff")})
    for kind, size, seed, mbl, lvl, wb, st, cap in (
            ("table", 140000, 3, 10000, 6, 15, 0, None), ("bitmap", 200000, 4, 32768, 9, 31, 0, None),
            ("random", 70000, 5, 4096, 6, 15, 0, None), ("zero", 100000, 6, 1000, 6, -15, 0, None),
            ("runs", 90000, 7, 7777, 1, 15, 3, None), ("text", 50000, 8, 300, 3, 15, 0, None),
            ("object", 120000, 9, 65536, 4, 31, 1, None), ("text", 80000, 10, 5000, 6, 15, 2, None),
            ("text", 80000, 11, 5000, 6, 15, 0, 20000), ("random", 30000, 12, 100, 6, 31, 0, None)):
        d4 = corpus.make_buffer(kind, size, seed)
        rc, out = R.compress(d4, lvl, window_bits=wb, strategy=st, max_block_len=mbl, dest_cap=cap)
        sections.append({"kind": kind, "size": size, "seed": seed, "max_block_len": mbl, "level": lvl,
                         "window_bits": wb, "strategy": st, "dest_cap": cap, "rc": rc,
                         "out_len": len(out), "out_sha256": sha(out),
                         "markers": out.count(b"\x00\x00\xff\xff"),
                         "boundaries": (size - 1) // mbl})
    # checksums
    sums = []
    for n in (0, 1, 15, 16, 17, 5551, 5552, 5553, 11105, 65536, 150001):
        d = corpus.make_buffer("random", n, n + 2)
        sums.append({"size": n, "seed": n + 2, "adler32": R.adler32(d), "crc32": R.crc32(d)})
    # inflate known answers
    kats = []
    for hx, wb, cap in INFCOVER:
        raw = bytes(int(x, 16) for x in hx.split())
        rc, out, used = R.uncompress(raw, cap, wb)
        kats.append({"hex": hx, "window_bits": wb, "dest_cap": cap, "rc": rc, "out_hex": out.hex(),
                     "consumed": used})
    # corrupted / truncated streams of a real buffer
    corrupt = []
    src = corpus.make_buffer("text", 20000, 3)
    rc, good = R.compress(src, 6)
    for pos in (0, 1, 2, 10, len(good) // 2, len(good) - 5, len(good) - 1):
        bad = bytearray(good)
        bad[pos] = (bad[pos] + 1) & 0xff
        rc, out, used = R.uncompress(bytes(bad), len(src), 15)
        corrupt.append({"flip": pos, "rc": rc, "out_len": len(out), "consumed": used,
                        "out_sha256": sha(out)})
    for cut in (1, 2, 7, len(good) // 2, len(good) - 4, len(good) - 1):
        rc, out, used = R.uncompress(good[:cut], len(src), 15)
        corrupt.append({"cut": cut, "rc": rc, "out_len": len(out), "consumed": used,
                        "out_sha256": sha(out)})
    # streams with full-flush markers between sections (the reference's own multi-section
    # output), damaged in seeded ways: what zsc_uncompress recovers after inflateSync
    import random
    rnd = random.Random(20)
    resync = []
    for kind, n, mbl, wb, lvl in (("text", 6000, 1000, 15, 6), ("zero", 5000, 700, 15, 6),
                                  ("table", 8000, 1500, 31, 9), ("runs", 4000, 512, -15, 1),
                                  ("object", 7000, 2000, 31, 6), ("bitmap", 9000, 3000, 15, 4)):
        data = corpus.make_buffer(kind, n, 77)
        rc, comp = R.compress(data, lvl, window_bits=wb, max_block_len=mbl, dest_cap=2 * n + 1000)
        assert rc == 0 and comp.count(b"\x00\x00\xff\xff") >= 1
        cases = []
        for k in range(28):
            ops = []  # edits applied in order: ["x", pos, hex] overwrite, ["d", pos, count] delete, ["t", len] truncate
            ln = len(comp)
            mode = k % 6
            if mode == 0:
                for _ in range(rnd.randrange(1, 4)):
                    i = rnd.randrange(ln)
                    ops.append(["f", i, 1 << rnd.randrange(8)])
            elif mode == 1:
                ops.append(["x", rnd.randrange(ln), "%02x" % rnd.randrange(256)])
            elif mode == 2:
                i = rnd.randrange(ln)
                j = min(ln, i + rnd.randrange(1, 40))
                ops.append(["x", i, bytes(rnd.randrange(256) for _ in range(j - i)).hex()])
            elif mode == 3:
                ops.append(["d", rnd.randrange(ln), rnd.randrange(1, 20)])
            elif mode == 4:
                ln = rnd.randrange(1, ln + 1)
                ops.append(["t", ln])
                if ln > 4:
                    ops.append(["f", rnd.randrange(ln), 0x55])
            else:
                ops.append(["f", rnd.randrange(min(ln, 12)), 1 << rnd.randrange(8)])
            b = apply_edits(comp, ops)
            cap = (n, n + 100, max(1, n // 2))[k % 3]
            rc, out, used = R.uncompress(bytes(b), cap, wb)
            cases.append({"edits": ops, "dest_cap": cap, "rc": rc, "out_len": len(out),
                          "consumed": used, "out_sha256": sha(out)})
        resync.append({"kind": kind, "size": n, "seed": 77, "window_bits": wb, "stream_hex": comp.hex(),
                       "cases": cases})
    json.dump({"deflate": deflate_cases, "streams": streams, "params": params, "small": small,
               "sections": sections, "strategies": strategies,
               "gz_header": gzh, "gz_header_misc": gz_misc, "gz_header_sections": gz_sections, "stored": stored}, open(os.path.join(HERE, "deflate_golden.json"), "w"), indent=0)
    json.dump({"checksums": sums, "inflate_kat": kats, "corrupt": corrupt, "resync": resync,
               "corrupt_source": {"kind": "text", "size": 20000, "seed": 3, "level": 6}},
              open(os.path.join(HERE, "inflate_golden.json"), "w"), indent=0)
    print(len(deflate_cases), "deflate cases,", len(kats), "inflate KATs")


if __name__ == "__main__":
    main()
