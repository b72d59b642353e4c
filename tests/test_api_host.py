"""The C-ABI library on a machine without a GPU: it loads, exports every symbol the
headers declare, and the host-only parts (sizing arithmetic, argument validation)
behave like the reference.  No kernel is launched here."""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G_DEF = json.load(open(os.path.join(ROOT, "tests", "golden", "deflate_golden.json")))


def declared_symbols():
    names = set()
    for hdr in ("include/zsc/zsc_pub.h", "include/zsc_hip.h"):
        text = open(os.path.join(ROOT, hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(zsc_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_loads_and_exports_every_declared_symbol():
    import zsc_amd
    syms = declared_symbols()
    assert len([s for s in syms if not s.startswith("zsc_hip_")]) == 16  # reference zsc_pub.h:86-411
    for name in syms:
        assert hasattr(zsc_amd.lib, name), f"{name} declared in include/ but not exported"


def test_product_does_not_reference_the_oracle():
    """The shipped sources never include, link or import anything under oracle/ or tests/."""
    for root, _, files in os.walk(os.path.join(ROOT, "zsc_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_py" not in text and "zsc_oracle" not in text and "libzsc_emu" not in text, f
                assert "import zlib" not in text, f


def test_sizing_helpers_match_oracle_and_reference(oracle):
    import zsc_amd
    assert zsc_amd.compress_get_min_work_buf_size() == (0, 333600)   # SURVEY 8a21, LP64 reference
    assert zsc_amd.uncompress_get_min_work_buf_size() == (0, 39920)
    for n in (0, 1, 100, 65536, 1029744, 1 << 26):
        for mbl in (1, 100, 20000, 1 << 20):
            for lvl in (0, 1, 6, 9):
                for wb, ml in ((15, 8), (-15, 8), (31, 8), (12, 8), (15, 5), (8, 8), (-8, 8), (7, 8), (15, 0)):
                    assert zsc_amd.compress_get_max_output_size2(n, mbl, lvl, wb, ml) == \
                           oracle.max_output(n, mbl, lvl, wb, ml), (n, mbl, lvl, wb, ml)


def test_sizing_helpers_vs_reference(reference):
    import zsc_amd
    L = reference.lib
    for wb in (15, -15, 31, 9, 8, -8, 7, 16, 24, 0, 47):
        for ml in (0, 1, 8, 9, 10):
            a = C.c_uint32()
            rc = L.zsc_compress_get_min_work_buf_size2(wb, ml, C.byref(a))
            assert zsc_amd.compress_get_min_work_buf_size(wb, ml) == (rc, a.value), (wb, ml)
        a = C.c_uint32(12345)
        rc = L.zsc_uncompress_get_min_work_buf_size2(wb, C.byref(a))
        got = zsc_amd.uncompress_get_min_work_buf_size(wb)
        assert got[0] == rc and (rc != 0 or got[1] == a.value), wb


def test_argument_validation_needs_no_gpu():
    """Errors the reference reports before it touches the data (src/zsc_compress.c:74-117)."""
    import zsc_amd
    data = b"x" * 1000
    assert zsc_amd.compress(data, work_len=333599)[0] == zsc_amd.Z_MEM_ERROR
    assert zsc_amd.compress2(data, window_bits=7)[0] == zsc_amd.Z_STREAM_ERROR
    assert zsc_amd.compress2(data, mem_level=10)[0] == zsc_amd.Z_STREAM_ERROR
    assert zsc_amd.compress2(data, level=10)[0] == zsc_amd.Z_STREAM_ERROR
    assert zsc_amd.compress2(data, strategy=5)[0] == zsc_amd.Z_STREAM_ERROR
    assert zsc_amd.compress2(data, window_bits=-8)[0] == zsc_amd.Z_STREAM_ERROR
    assert zsc_amd.uncompress(b"\x78\x9c", 10, work_len=100)[0] == zsc_amd.Z_MEM_ERROR
    for c in G_DEF["params"]:
        if c["rc"] != 0:
            assert zsc_amd.compress2(data, level=c["level"], window_bits=c["window_bits"],
                                     mem_level=c["mem_level"], strategy=c["strategy"],
                                     work_len=333600)[0] == c["rc"], c


def test_gz_header_argument_errors_need_no_gpu():
    """a gz_header on a zlib-wrapped call is Z_STREAM_ERROR (deflateSetHeader / inflateGetHeader)"""
    import zsc_amd
    data = b"y" * 500
    m = G_DEF["gz_header_misc"]
    h, keep = zsc_amd.gz_header_for_writing(name=b"n")
    assert zsc_amd.compress2(data, window_bits=15, gz_header=h)[0] == m["header_on_zlib_compress"] == -2
    hr, bufs = zsc_amd.gz_header_for_reading(4, 4, 4)
    assert zsc_amd.uncompress2(b"\x78\x9c\x03\x00\x00\x00\x00\x01", 10, 15, gz_header=hr)[0] == \
        m["header_on_zlib_uncompress"] == -2


def test_null_pointers_die_in_zsc_assert():
    """The reference's ZlibDeathTest.Asserts (test/zlib_gtest.cpp:2093-2396): a NULL pointer into a
    zsc_* function is not an error code, it trips ZSC_ASSERT naming the parameter
    (src/zsc_compress.c:56-59, src/zsc_uncompr.c:48-52); max_block_len == 0 too (:124).
    Each case in its own process; none of them gets as far as the GPU."""
    import subprocess
    import sys
    import zsc_amd
    prologue = (
        "import ctypes as C, sys\n"
        f"L = C.CDLL({zsc_amd.lib_path!r})\n"
        "n = C.c_uint32(0); m = C.c_uint32(100); s = C.c_uint32(8)\n"
        "src = C.create_string_buffer(b'abcdefgh'); dst = C.create_string_buffer(100); work = C.create_string_buffer(400000)\n"
        "p = C.byref\n")
    cases = [
        ("size_out", "L.zsc_compress_get_min_work_buf_size(None)"),
        ("size_out", "L.zsc_compress_get_min_work_buf_size2(15, 8, None)"),
        ("size_out", "L.zsc_compress_get_max_output_size(8, 8, 6, None)"),
        ("size_out", "L.zsc_compress_get_max_output_size2(8, 8, 6, 15, 8, None)"),
        ("size_out", "L.zsc_compress_get_max_output_size_gzip(8, 8, 6, None, None)"),
        ("size_out", "L.zsc_compress_get_max_output_size_gzip2(8, 8, 6, 31, 8, None, None)"),
        ("max_block_len", "L.zsc_compress_get_max_output_size(8, 0, 6, p(n))"),
        ("size_out", "L.zsc_uncompress_get_min_work_buf_size(None)"),
        ("size_out", "L.zsc_uncompress_get_min_work_buf_size2(15, None)"),
        ("dest", "L.zsc_compress(None, p(m), src, 8, 8, work, 400000, 6)"),
        ("dest_len", "L.zsc_compress(dst, None, src, 8, 8, work, 400000, 6)"),
        ("source", "L.zsc_compress(dst, p(m), None, 8, 8, work, 400000, 6)"),
        ("work", "L.zsc_compress(dst, p(m), src, 8, 8, None, 400000, 6)"),
        ("max_block_len", "L.zsc_compress(dst, p(m), src, 8, 0, work, 400000, 6)"),
        ("dest", "L.zsc_compress2(None, p(m), src, 8, 8, work, 400000, 6, 15, 8, 0)"),
        ("source", "L.zsc_compress_gzip(dst, p(m), None, 8, 8, work, 400000, 6, None)"),
        ("work", "L.zsc_compress_gzip2(dst, p(m), src, 8, 8, None, 400000, 6, 31, 8, 0, None)"),
        ("dest", "L.zsc_uncompress(None, p(m), src, p(s), work, 400000)"),
        ("dest_len", "L.zsc_uncompress(dst, None, src, p(s), work, 400000)"),
        ("source", "L.zsc_uncompress(dst, p(m), None, p(s), work, 400000)"),
        ("source_len", "L.zsc_uncompress(dst, p(m), src, None, work, 400000)"),
        ("work", "L.zsc_uncompress(dst, p(m), src, p(s), None, 400000)"),
        ("dest", "L.zsc_uncompress2(None, p(m), src, p(s), work, 400000, 15)"),
        ("source", "L.zsc_uncompress_gzip(dst, p(m), None, p(s), work, 400000, None)"),
        ("work", "L.zsc_uncompress_gzip2(dst, p(m), src, p(s), None, 400000, 31, None)"),
    ]
    procs = [(name, call, subprocess.Popen([sys.executable, "-c", prologue + call + "\nsys.exit(0)\n"],
                                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
             for name, call in cases]
    for name, call, pr in procs:
        err = pr.communicate()[1].decode(errors="replace")
        assert pr.returncode == -6 and "Assertion" in err and name in err, (call, pr.returncode, err[-300:])
