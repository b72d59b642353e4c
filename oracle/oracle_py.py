"""ctypes bindings for the checkers -- TEST INFRASTRUCTURE ONLY.

Two libraries live under oracle/:
  libzsc_oracle.so        our CPU restatement (zsc_oracle.c)           -> Oracle
  _ref/libzsc_ref.so      the reference itself, compiled by the Makefile -> Reference
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product (zsc_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))

Z_OK, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR = 0, -2, -3, -4, -5
COMPRESS_WORK = 333600  # reference zsc_compress_get_min_work_buf_size on LP64
UNCOMPRESS_WORK = 39920


def build(ref: bool = True) -> None:
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"] + (["ref"] if ref else []), check=True)


class Symbol(C.Structure):
    _fields_ = [("dist", C.c_uint16), ("lc", C.c_uint8), ("pad", C.c_uint8)]


class Block(C.Structure):
    _fields_ = [("sym_begin", C.c_uint32), ("sym_count", C.c_uint32), ("in_begin", C.c_uint32),
                ("in_len", C.c_uint32), ("stored_ok", C.c_uint8), ("last", C.c_uint8),
                ("pad", C.c_uint8 * 2)]


class Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "libzsc_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.zo_adler32.restype = C.c_uint32
        L.zo_adler32.argtypes = [C.c_uint32, C.c_char_p, C.c_uint32]
        L.zo_crc32.restype = C.c_uint32
        L.zo_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_uint32]
        L.zo_compress.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_int)]
        L.zo_uncompress.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_char_p,
                                    C.POINTER(C.c_uint32), C.c_uint32, C.c_int]
        L.zo_compress_max_output.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_uint32)]
        L.zo_parse.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p,
                               C.POINTER(C.c_uint32)]

    def adler32(self, data: bytes, start: int = 1) -> int:
        return self.lib.zo_adler32(start, data, len(data))

    def crc32(self, data: bytes, start: int = 0) -> int:
        return self.lib.zo_crc32(start, data, len(data))

    def max_output(self, n, max_block_len, level=6, window_bits=15, mem_level=8):
        out = C.c_uint32()
        rc = self.lib.zo_compress_max_output(n, max_block_len, level, window_bits, mem_level,
                                             C.byref(out))
        return rc, out.value

    def compress(self, data: bytes, level=6, window_bits=15, mem_level=8, strategy=0,
                 max_block_len=None, dest_cap=None, work_len=COMPRESS_WORK):
        """-> (rc, bytes, unsupported)"""
        n = len(data)
        mbl = max(n, 1) if max_block_len is None else max_block_len
        if dest_cap is None:
            rc, dest_cap = self.max_output(n, mbl, level, window_bits, mem_level)
            if rc != 0:
                dest_cap = n + (n >> 3) + 128
        dst = C.create_string_buffer(max(dest_cap, 1))
        dl = C.c_uint32(dest_cap)
        uns = C.c_int(0)
        rc = self.lib.zo_compress(dst, C.byref(dl), data, n, mbl, work_len, level, window_bits,
                                  mem_level, strategy, C.byref(uns))
        return rc, dst.raw[:dl.value], bool(uns.value)

    def uncompress(self, data: bytes, dest_cap: int, window_bits=15, work_len=UNCOMPRESS_WORK):
        """-> (rc, bytes, consumed)"""
        dst = C.create_string_buffer(max(dest_cap, 1))
        dl = C.c_uint32(dest_cap)
        sl = C.c_uint32(len(data))
        rc = self.lib.zo_uncompress(dst, C.byref(dl), data, C.byref(sl), work_len, window_bits)
        return rc, dst.raw[:dl.value], sl.value

    def parse(self, data: bytes, level=6, wbits=15, mem_level=8, strategy=0):
        """-> (symbols as list of (dist, lc), blocks as list of dict)"""
        n = len(data)
        syms = (Symbol * (n + 1))()
        blocks = (Block * (n // ((1 << (mem_level + 6)) - 1) + 2))()
        ns, nb = C.c_uint32(), C.c_uint32()
        rc = self.lib.zo_parse(data, n, level, wbits, mem_level, strategy, syms, C.byref(ns),
                               blocks, C.byref(nb))
        assert rc == 0, rc
        return syms, ns.value, blocks, nb.value


class Reference:
    """The compiled reference (only where oracle/_ref/libzsc_ref.so exists)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(os.path.join(_HERE, "_ref", "libzsc_ref.so"))

    def __init__(self):
        self.lib = L = C.CDLL(os.path.join(_HERE, "_ref", "libzsc_ref.so"))
        L.zref_adler32.restype = C.c_uint32
        L.zref_adler32.argtypes = [C.c_uint32, C.c_char_p, C.c_uint32]
        L.zref_crc32.restype = C.c_uint32
        L.zref_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_uint32]
        L.zsc_compress_gzip2.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32,
                                         C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]
        L.zsc_uncompress_gzip2.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_char_p,
                                           C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_int,
                                           C.c_void_p]
        L.zsc_compress_get_max_output_size2.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                        C.c_int, C.POINTER(C.c_uint32)]
        self._work = C.create_string_buffer(400000)

    def adler32(self, data: bytes, start: int = 1) -> int:
        return self.lib.zref_adler32(start, data, len(data))

    def crc32(self, data: bytes, start: int = 0) -> int:
        return self.lib.zref_crc32(start, data, len(data))

    def max_output(self, n, max_block_len, level=6, window_bits=15, mem_level=8):
        out = C.c_uint32()
        rc = self.lib.zsc_compress_get_max_output_size2(n, max_block_len, level, window_bits,
                                                        mem_level, C.byref(out))
        return rc, out.value

    def compress(self, data: bytes, level=6, window_bits=15, mem_level=8, strategy=0,
                 max_block_len=None, dest_cap=None, work_len=COMPRESS_WORK, gz_header=None):
        """gz_header: a ctypes struct laid out like the reference's gz_header (passed by reference)"""
        n = len(data)
        mbl = max(n, 1) if max_block_len is None else max_block_len
        if dest_cap is None:
            rc, dest_cap = self.max_output(n, mbl, level, window_bits, mem_level)
            if rc != 0:
                dest_cap = n + (n >> 3) + 128
            if gz_header is not None:
                dest_cap += 70000 * 3
        dst = C.create_string_buffer(max(dest_cap, 1))
        dl = C.c_uint32(dest_cap)
        rc = self.lib.zsc_compress_gzip2(dst, C.byref(dl), data, n, mbl, self._work, work_len,
                                         level, window_bits, mem_level, strategy,
                                         None if gz_header is None else C.byref(gz_header))
        return rc, dst.raw[:dl.value]

    def uncompress(self, data: bytes, dest_cap: int, window_bits=15, work_len=UNCOMPRESS_WORK, gz_header=None):
        dst = C.create_string_buffer(max(dest_cap, 1))
        dl = C.c_uint32(dest_cap)
        sl = C.c_uint32(len(data))
        rc = self.lib.zsc_uncompress_gzip2(dst, C.byref(dl), data, C.byref(sl), self._work,
                                           work_len, window_bits,
                                           None if gz_header is None else C.byref(gz_header))
        return rc, dst.raw[:dl.value], sl.value
