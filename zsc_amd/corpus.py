"""Seeded synthetic workloads for the DEFLATE hot path (no network, no corpora).

The reference benchmarks on the Canterbury corpus, which its build fetches with
git (reference CMakeLists.txt:104-174) and which is not available offline.
SURVEY.md section 8d therefore defines a *Canterbury-like* set: 11 buffers with
the real file sizes (order of reference test/zlib_gtest.cpp:113-125) and
contents generated from fixed seeds so that every class of behaviour the real
files trigger is present:

  text    Zipf-distributed words, punctuation, ~70 column lines
          (alice29 / lcet10 / plrabn12 / asyoulik)
  token   small-vocabulary program / markup text with indentation runs
          (fields.c / cp.html / grammar.lsp / xargs.1)
  bitmap  sparse bi-level raster, 216 byte rows, strong vertical correlation,
          long zero runs (ptt5) -- thousands of same-hash candidates per position
  table   fixed-size binary records with slowly varying fields (kennedy.xls)
  object  opcode-like words, zero padding, a string table (sum)

All randomness comes from a counter-based splitmix64 evaluated with numpy
uint64 arithmetic, so a (kind, size, seed) triple is the same bytes on every
machine and numpy version.  Nothing here is timed; it only makes inputs.
"""
from __future__ import annotations

import numpy as np

# name, size in bytes, generator kind -- order of reference test/zlib_gtest.cpp:113-125
CANTERBURY_LIKE = (
    ("alice29.txt", 152089, "text"),
    ("ptt5", 513216, "bitmap"),
    ("fields.c", 11150, "token"),
    ("kennedy.xls", 1029744, "table"),
    ("sum", 38240, "object"),
    ("lcet10.txt", 426754, "text"),
    ("plrabn12.txt", 481861, "text"),
    ("cp.html", 24603, "token"),
    ("grammar.lsp", 3721, "token"),
    ("xargs.1", 4227, "token"),
    ("asyoulik.txt", 125179, "text"),
)
CANTERBURY_TOTAL = sum(s for _, s, _ in CANTERBURY_LIKE)  # 2 810 784, reference README.md:143

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


class Stream:
    """Counter-based random stream: draw k is mix(seed * C + k)."""

    def __init__(self, seed: int, lane: int = 0):
        with np.errstate(over="ignore"):
            self._key = _mix(np.array([(seed * 0x632BE59BD9B4E019 + lane * 0x1234567) & 0xFFFFFFFFFFFFFFFF],
                                      dtype=np.uint64))[0]
        self._ctr = 0

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(self._ctr, self._ctr + n, dtype=np.uint64)
            self._ctr += n
            return _mix(idx * np.uint64(0xD1342543DE82EF95) + self._key)

    def below(self, n: int, bound: int) -> np.ndarray:
        """n integers uniform in [0, bound) (bound < 2**32)."""
        return ((self.u64(n) >> np.uint64(32)) * np.uint64(bound) >> np.uint64(32)).astype(np.int64)

    def bytes(self, n: int) -> np.ndarray:
        words = self.u64((n + 7) // 8)
        return words.view(np.uint8)[:n].copy()


def _zipf_picks(st: Stream, n: int, vocab: int, s_num: int = 11, s_den: int = 10) -> np.ndarray:
    """n ranks in [0, vocab) with P(r) ~ 1/(r+1)^(s_num/s_den), integer CDF."""
    w = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s_num / s_den)
    cdf = np.floor(np.cumsum(w) / w.sum() * float(1 << 32)).astype(np.uint64)
    cdf[-1] = np.uint64(1 << 32)
    u = st.u64(n) >> np.uint64(32)
    return np.searchsorted(cdf, u, side="right").astype(np.int64).clip(0, vocab - 1)


def _make_vocab(st: Stream, count: int, alphabet: bytes, min_len: int, max_len: int):
    lens = min_len + st.below(count, max_len - min_len + 1)
    # frequent words are short
    lens = np.sort(lens)
    letters = np.frombuffer(alphabet, dtype=np.uint8)
    # letter frequencies skewed the way English is (first letters of `alphabet` common)
    pick = _zipf_picks(st, int(lens.sum()), len(letters), 7, 10)
    flat = letters[pick]
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    return flat, starts, lens


def _join_words(flat, starts, lens, order, seps) -> np.ndarray:
    """Concatenate words[order[i]] + seps[i] without a Python loop."""
    wl = lens[order]
    total = int(wl.sum() + len(order))
    out = np.empty(total, dtype=np.uint8)
    ends = np.cumsum(wl + 1)
    begins = ends - (wl + 1)
    # index of each output byte inside its word
    word_of = np.repeat(np.arange(len(order)), wl + 1)
    off = np.arange(total) - begins[word_of]
    is_sep = off == wl[word_of]
    src = starts[order][word_of] + np.minimum(off, wl[word_of] - 1)
    out[:] = flat[src]
    out[is_sep] = seps[word_of[is_sep]]
    return out


def gen_text(size: int, seed: int) -> bytes:
    st = Stream(seed, 1)
    flat, starts, lens = _make_vocab(st, 6000, b"etaoinshrdlcumwfgypbvkjxqz", 2, 10)
    nwords = size // 2 + 64
    order = _zipf_picks(st, nwords, 6000)
    r = st.below(nwords, 100)
    seps = np.full(nwords, ord(" "), dtype=np.uint8)
    seps[r < 8] = ord(",")
    seps[r < 4] = ord(".")
    seps[(r >= 8) & (r < 20)] = ord("\n")
    body = _join_words(flat, starts, lens, order, seps)
    # capitalise after full stops
    dots = np.nonzero(body[:-2] == ord("."))[0]
    nxt = dots + 1
    ok = (body[nxt] >= ord("a")) & (body[nxt] <= ord("z"))
    body[nxt[ok]] -= 32
    return body[:size].tobytes()


def gen_token(size: int, seed: int) -> bytes:
    st = Stream(seed, 2)
    flat, starts, lens = _make_vocab(st, 300, b"etrinasoldcpumfh_gbvwyxkqjz0123456789", 2, 9)
    nwords = size // 2 + 64
    order = _zipf_picks(st, nwords, 300, 9, 10)
    r = st.below(nwords, 100)
    punct = np.frombuffer(b" (){};=,<>/\"*-+&|![]\n\t", dtype=np.uint8)
    seps = punct[np.minimum(r // 5, len(punct) - 1)]
    seps[r >= 60] = ord(" ")
    body = _join_words(flat, starts, lens, order, seps)
    # indentation: after a newline insert a run of spaces by overwriting
    nl = np.nonzero(body[:-12] == ord("\n"))[0]
    depth = st.below(len(nl), 3) * 4
    for d in (4, 8):
        sel = nl[depth >= d]
        for k in range(d - 3, d + 1):
            body[sel + k] = ord(" ")
    return body[:size].tobytes()


def gen_bitmap(size: int, seed: int) -> bytes:
    st = Stream(seed, 3)
    row_bytes = 216  # 1728 pixels, the CCITT fax width ptt5 uses
    rows = size // row_bytes + 2
    img = np.zeros((rows, row_bytes), dtype=np.uint8)
    # a few dozen "strokes": vertical-ish bands that persist over many rows
    nstroke = 60
    col = st.below(nstroke, row_bytes)
    top = st.below(nstroke, rows)
    height = 20 + st.below(nstroke, 400)
    width = 1 + st.below(nstroke, 6)
    pat = st.bytes(nstroke)
    for i in range(nstroke):
        r0, r1 = int(top[i]), min(rows, int(top[i] + height[i]))
        c0, c1 = int(col[i]), min(row_bytes, int(col[i] + width[i]))
        img[r0:r1, c0:c1] = pat[i] | 0x81
    # text-like speckle lines: every ~30 rows a band of 12 rows with sparse marks
    band = st.below(rows // 30 + 1, 30)
    for b, off in enumerate(band):
        r0 = b * 30 + int(off)
        if r0 + 12 >= rows:
            break
        marks = st.below(40, row_bytes)
        vals = st.bytes(40)
        jitter = st.below(12 * 40, 100).reshape(12, 40)
        for k in range(12):
            keep = jitter[k] < 70
            img[r0 + k, marks[keep]] = vals[keep]
    return img.reshape(-1)[:size].tobytes()


def gen_table(size: int, seed: int) -> bytes:
    st = Stream(seed, 4)
    rec = 29  # odd record size so fields drift against dword alignment
    n = size // rec + 2
    t = np.zeros((n, rec), dtype=np.uint8)
    t[:, 0] = 0x03
    t[:, 1] = 0x02
    t[:, 2] = 0x0E
    idx = np.arange(n, dtype=np.uint32)
    t[:, 4] = (idx // 17) & 0xFF
    t[:, 5] = (idx // 17) >> 8
    t[:, 6] = idx % 17
    t[:, 8] = 0x0F + (st.below(n, 8) == 0) * 3
    # an IEEE double whose low mantissa bytes are noisy and high bytes repeat
    t[:, 10:13] = st.bytes(n * 3).reshape(n, 3)
    hi = st.below(n, 5)
    t[:, 13] = (hi * 37) & 0xFF
    t[:, 14] = 0x40 + hi
    t[:, 15] = 0x40
    # trailing style block identical in most records
    t[:, 16:29] = np.frombuffer(b"\x00\x00\x15\x00\xff\x00\x00\x00\x01\x00\x20\x00\x00", dtype=np.uint8)
    odd = st.below(n, 13) == 0
    t[odd, 18] = 0x16
    return t.reshape(-1)[:size].tobytes()


def gen_object(size: int, seed: int) -> bytes:
    st = Stream(seed, 5)
    nwords = size // 4 + 2
    ops = np.array([0x9DE3BF98, 0x01000000, 0x81C7E008, 0x81E80000, 0x40000000, 0xD0072044,
                    0x90102000, 0x80A22000, 0x12800005, 0xC2002000, 0x92102001, 0x7FFFFF00],
                   dtype=np.uint32)
    pick = _zipf_picks(st, nwords, len(ops), 8, 10)
    w = ops[pick].copy()
    noise = st.below(nwords, 1 << 13).astype(np.uint32)
    has_imm = st.below(nwords, 3) != 0
    w[has_imm] |= noise[has_imm]
    body = w.byteswap().view(np.uint8).copy()  # big-endian words, like SPARC
    # zero padding runs and a string table in the last fifth
    cut = (size * 4 // 5) & ~3
    pad_at = st.below(6, max(cut - 600, 1))
    for p in pad_at:
        body[int(p):int(p) + 512] = 0
    tail = np.frombuffer(gen_token(size - cut + 16, seed ^ 0x55), dtype=np.uint8).copy()
    tail[tail == ord(" ")] = 0
    body[cut:size] = tail[: size - cut]
    return body[:size].tobytes()


_GEN = {"text": gen_text, "token": gen_token, "bitmap": gen_bitmap, "table": gen_table,
        "object": gen_object}


def make_buffer(kind: str, size: int, seed: int) -> bytes:
    """One synthetic buffer.  kind: text|token|bitmap|table|object|random|zero|runs."""
    if size == 0:
        return b""
    if kind == "random":
        return Stream(seed, 9).bytes(size).tobytes()
    if kind == "zero":
        return bytes(size)
    if kind == "runs":  # run-heavy binary: short repeated patterns of random length
        st = Stream(seed, 7)
        n = size // 8 + 8
        vals = st.below(n, 6).astype(np.uint8) * 41
        lens = 1 + st.below(n, 40)
        return np.repeat(vals, lens)[:size].tobytes().ljust(size, b"\0")
    out = _GEN[kind](size, seed)
    assert len(out) == size, (kind, size, len(out))
    return out


def canterbury_like(seed: int = 0):
    """The 11-buffer Canterbury-like set for one seed: list of (name, bytes)."""
    return [(name, make_buffer(kind, size, seed * 131 + i))
            for i, (name, size, kind) in enumerate(CANTERBURY_LIKE)]


def mix64k(count: int, seed: int = 0):
    """BASELINE config 3: `count` buffers of 64 KiB, one third random, zero, text."""
    out = []
    for i in range(count):
        kind = ("random", "zero", "text")[i % 3]
        out.append(make_buffer(kind, 65536, seed * 7919 + i))
    return out
