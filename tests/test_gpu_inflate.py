"""Parity of the HIP inflate path with the oracle / reference golden vectors (MI355X)."""
import hashlib
import json
import os
import zlib as stock_zlib  # only to MAKE foreign streams (another encoder); never a checker

import pytest

pytestmark = pytest.mark.gpu

from zsc_amd import corpus  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
G_INF = json.load(open(os.path.join(HERE, "golden", "inflate_golden.json")))


@pytest.fixture(scope="module")
def z():
    import zsc_amd
    assert zsc_amd.lib.zsc_hip_init(-1) == 0
    return zsc_amd


def test_reference_known_answers(z):
    """Inputs of the reference's test/infcover.c, outcomes recorded from the reference build."""
    for c in G_INF["inflate_kat"]:
        raw = bytes(int(x, 16) for x in c["hex"].split())
        rc, out, used = z.uncompress2(raw, c["dest_cap"], c["window_bits"])
        assert (rc, out.hex(), used) == (c["rc"], c["out_hex"], c["consumed"]), c


def test_roundtrip_truncation_small_dest_batch(z, oracle):
    cases = []
    for n in (0, 1, 5, 258, 259, 4096, 40000, 65536, 150000):
        for kind in ("text", "bitmap", "table", "random", "zero", "runs"):
            data = corpus.make_buffer(kind, n, n + 3)
            for lvl in (1, 6, 9):
                comp = oracle.compress(data, lvl)[1]
                cases.append((comp, n))
                cases.append((comp + b"junk", n + 10))
                if n:
                    cases.append((comp, n - 1))
                    cases.append((comp[:len(comp) // 2], n))
                    cases.append((comp[:len(comp) - 3], n))
            for lvl in (0, 6):
                cases.append((stock_zlib.compress(data, lvl), n))
    rc, outs, used, stats = z.uncompress_batch([c for c, _ in cases], [cap for _, cap in cases])
    assert rc == 0
    for (comp, cap), o, u, s in zip(cases, outs, used, stats):
        assert (s, o, u) == oracle.uncompress(comp, cap), (len(comp), cap)


def test_wrappers_gzip_raw(z, oracle):
    data = corpus.make_buffer("text", 50000, 9)
    for wb in (31, -15):
        comp = oracle.compress(data, 6, window_bits=wb)[1]
        assert z.uncompress2(comp, len(data), wb) == (0, data, len(comp))
        assert z.uncompress2(comp, len(data), wb) == oracle.uncompress(comp, len(data), wb)
    gz = oracle.compress(data, 6, window_bits=31)[1]
    assert z.uncompress_gzip(gz, len(data)) == (0, data, len(gz))
    bad = bytearray(gz)
    bad[-6] ^= 1                                        # CRC-32 trailer
    assert z.uncompress_gzip(bytes(bad), len(data))[0] == -3
    assert z.uncompress(b"\x78\x9c", 10, work_len=100)[0] == -4  # Z_MEM_ERROR before any decoding


def test_deflate_then_inflate_on_gpu_is_identity(z):
    bufs = [b for _, b in corpus.canterbury_like(2)]
    rc, comps, stats = z.compress_batch(bufs, level=6)
    assert rc == 0 and all(s == 0 for s in stats)
    rc, outs, used, stats = z.uncompress_batch(comps, [len(b) for b in bufs])
    assert rc == 0
    for b, c, o, u, s in zip(bufs, comps, outs, used, stats):
        assert (s, o, u) == (0, b, len(c))


def test_config4_many_gzip_chunks(z, oracle):
    """BASELINE config 4 (scaled to what the CPU checker can prepare): thousands of
    independent gzip members of 4..64 KiB through one batched inflate call."""
    import random
    rnd = random.Random(4)
    kinds = ("text", "token", "table", "bitmap", "object", "random", "zero")
    plain = [corpus.make_buffer(kinds[i % len(kinds)], rnd.randrange(4096, 65537), 100 + i) for i in range(160)]
    gz = [oracle.compress(b, (1, 6, 9)[i % 3], window_bits=31)[1] for i, b in enumerate(plain)]
    reps = 16                                     # 2560 streams
    rc, outs, _, stats = z.uncompress_batch(gz * reps, [len(b) for b in plain] * reps, window_bits=31)
    assert rc == 0 and all(s == 0 for s in stats)
    assert outs == plain * reps


def test_more_streams_than_the_grid_holds_and_plan_alignment(z, oracle):
    """The decoder's groups take streams from one queue: with more streams than resident groups
    (256 CUs x 20 waves x 4) every group decodes several, of different kinds and lengths, one
    after the other in the same LDS.  And: a plan refuses offsets that are not multiples of 16."""
    import ctypes as C
    kinds = ("text", "token", "table", "zero", "random")
    plain = [corpus.make_buffer(kinds[i % len(kinds)], 1 + (i * 37) % 700, 900 + i) for i in range(48)]
    gz = [oracle.compress(b, (1, 6, 9)[i % 3], window_bits=(15, 31, -15)[i % 3])[1] for i, b in enumerate(plain)]
    for wb in (15, 31, -15):
        idx = [i for i in range(48) if (15, 31, -15)[i % 3] == wb]
        reps = 100000 // len(idx) + 1                  # > 81 920 resident groups
        srcs = [gz[i] for i in idx] * reps
        caps = [len(plain[i]) for i in idx] * reps
        rc, outs, _, stats = z.uncompress_batch(srcs, caps, window_bits=wb)
        assert rc == 0 and all(s == 0 for s in stats)
        want = [plain[i] for i in idx]
        assert all(outs[k] == want[k % len(idx)] for k in range(len(outs)))
    h = C.c_void_p()
    rc = z.lib.zsc_hip_inflate_plan_create(C.byref(h), 2, (C.c_uint32 * 2)(10, 10), (C.c_uint64 * 2)(0, 88),
                                           (C.c_uint32 * 2)(100, 100), (C.c_uint64 * 2)(0, 176), 15)
    assert rc == -2 and not h.value


def test_resynchronisation_after_data_errors(z, oracle):
    """damaged multi-section streams written by the reference; zsc_uncompress
    finds the next full-flush marker (inflateSync) and salvages what follows.  Expected code,
    bytes and consumed count were recorded from the reference; one batched call."""
    from test_oracle import apply_edits
    srcs, caps, want = [], [], []
    for r in G_INF["resync"]:
        stream = bytes.fromhex(r["stream_hex"])
        data = corpus.make_buffer(r["kind"], r["size"], r["seed"])
        assert z.uncompress2(stream, len(data), r["window_bits"]) == (0, data, len(stream))
        per_wb = [], [], []
        for c in r["cases"]:
            per_wb[0].append(apply_edits(stream, c["edits"]))
            per_wb[1].append(c["dest_cap"])
            per_wb[2].append(c)
        rc, outs, used, stats = z.uncompress_batch(per_wb[0], per_wb[1], window_bits=r["window_bits"])
        assert rc == 0
        for c, o, u, st in zip(per_wb[2], outs, used, stats):
            assert (st, len(o), u, hashlib.sha256(o).hexdigest()) == \
                   (c["rc"], c["out_len"], c["consumed"], c["out_sha256"]), c
    # seeded damage, kernel == oracle (the oracle is pinned to the reference on 40 000 such cases)
    import random
    rnd = random.Random(31)
    for wb in (15, 31, -15):
        bad, caps = [], []
        for i in range(400):
            n = rnd.choice([200, 3000, 20000, 70000])
            data = corpus.make_buffer(("text", "zero", "table", "random", "runs", "bitmap")[i % 6], n, i)
            comp = bytearray(oracle.compress(data, (1, 6, 9)[i % 3], window_bits=wb)[1])
            for _ in range(rnd.randrange(1, 3)):
                comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
            if i % 7 == 0:
                del comp[rnd.randrange(len(comp)):]
            bad.append(bytes(comp))
            caps.append(rnd.choice([n, n + 50, max(1, n // 2)]))
        rc, outs, used, stats = z.uncompress_batch(bad, caps, window_bits=wb)
        assert rc == 0
        for b, cap, o, u, st in zip(bad, caps, outs, used, stats):
            assert (st, o, u) == oracle.uncompress(b, cap, wb), (wb, cap, len(b))
