"""ctypes binding of libzsc_hip.so -- the host-side mirror of the reference API.

Function names follow ``include/zsc/zsc_pub.h`` with the ``zsc_`` prefix dropped;
arguments keep their reference meaning (``max_block_len``, caller-sized work buffer,
``window_bits`` wrapper encoding).  Return values are ``(ZlibReturn, bytes, ...)``
tuples instead of out-parameters.

The library is built in-tree by ``make -C zsc_amd/csrc`` (``__graft_entry__.build``)
and must exist: this module never falls back to a CPU codec.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

Z_OK, Z_STREAM_END, Z_NEED_DICT = 0, 1, 2
Z_ERRNO, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR, Z_VERSION_ERROR = -1, -2, -3, -4, -5, -6
Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED = 0, 1, 2, 3, 4
GZIP_CODE, DEF_WBITS, DEF_MEM_LEVEL = 16, 15, 8
NKERNELS = 9
KERNEL_NAMES = ("checksum", "hash_sort", "match_table", "parse", "parse_short", "huff_plan", "layout", "emit", "total")

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.environ.get("ZSC_HIP_LIB") or os.path.join(_HERE, "libzsc_hip.so")  # env: experiments only


def build_library() -> None:
    """Compile every HIP source for gfx950 into zsc_amd/libzsc_hip.so (hipcc, no GPU needed)."""
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc")], check=True)


def _load() -> C.CDLL:
    # PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, the same
    # as /opt/rocm's).  Two HIP runtimes in one process cannot both own the GPU, so when
    # torch is installed it is imported FIRST: the loader then resolves libzsc_hip.so's
    # NEEDED libamdhip64.so.7 to the copy torch already mapped, and tensors, streams and
    # our kernels share one runtime.  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C zsc_amd/csrc). zsc_amd has no CPU fallback.")
    L = C.CDLL(lib_path)
    u8p, u32p, i32p, u64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
    L.zsc_hip_init.argtypes = [C.c_int32]
    L.zsc_hip_device_info.restype = C.c_char_p
    L.zsc_compress_get_min_work_buf_size.argtypes = [u32p]
    L.zsc_compress_get_min_work_buf_size2.argtypes = [C.c_int32, C.c_int32, u32p]
    L.zsc_compress_get_max_output_size.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, u32p]
    L.zsc_compress_get_max_output_size2.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                                    C.c_int32, u32p]
    L.zsc_uncompress_get_min_work_buf_size.argtypes = [u32p]
    L.zsc_uncompress_get_min_work_buf_size2.argtypes = [C.c_int32, u32p]
    L.zsc_compress_gzip2.argtypes = [u8p, u32p, C.c_char_p, C.c_uint32, C.c_uint32, u8p, C.c_uint32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.zsc_uncompress_gzip2.argtypes = [u8p, u32p, C.c_char_p, u32p, u8p, C.c_uint32, C.c_int32,
                                       C.c_void_p]
    L.zsc_hip_compress_batch.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), u32p,
                                         C.POINTER(C.c_void_p), u32p, i32p, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32]
    L.zsc_hip_compress_sections_batch.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), u32p, u32p,
                                                  C.POINTER(C.c_void_p), u32p, i32p, C.c_int32,
                                                  C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
    L.zsc_hip_compress_sections_device.argtypes = [C.c_uint32, C.c_void_p, u64p, u32p, u32p, C.c_void_p,
                                                   u64p, u32p, u32p, i32p, C.c_int32, C.c_int32,
                                                   C.c_int32, C.c_int32]
    L.zsc_hip_uncompress_batch.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), u32p,
                                           C.POINTER(C.c_void_p), u32p, i32p, C.c_int32]
    L.zsc_hip_inflate_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, u32p, u64p, u32p,
                                              u64p, C.c_int32]
    L.zsc_hip_inflate_plan_create_ordered.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, u32p, u64p, u32p,
                                                      u64p, C.c_int32, u32p]
    L.zsc_hip_inflate_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.zsc_hip_inflate_plan_results.argtypes = [C.c_void_p, u32p, u32p, i32p, C.POINTER(C.c_float)]
    L.zsc_hip_inflate_plan_destroy.argtypes = [C.c_void_p]
    L.zsc_hip_inflate_plan_destroy.restype = None
    L.zsc_hip_deflate_plan_layout.argtypes = [C.c_uint32, u32p, C.c_int32, C.c_int32, C.c_int32,
                                              u64p, u64p, u32p, u64p, u64p]
    L.zsc_hip_deflate_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, u32p, u64p, u64p,
                                              u32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.zsc_hip_deflate_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.zsc_hip_deflate_plan_results.argtypes = [C.c_void_p, u32p, i32p]
    L.zsc_hip_deflate_plan_profile.argtypes = [C.c_void_p, C.c_int32]
    L.zsc_hip_deflate_plan_profile.restype = None
    L.zsc_hip_deflate_plan_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.zsc_hip_deflate_plan_scratch_bytes.argtypes = [C.c_void_p]
    L.zsc_hip_deflate_plan_scratch_bytes.restype = C.c_uint64
    L.zsc_hip_deflate_plan_sub_batches.argtypes = [C.c_void_p]
    L.zsc_hip_deflate_plan_sub_batches.restype = C.c_uint32
    L.zsc_hip_deflate_plan_destroy.argtypes = [C.c_void_p]
    L.zsc_hip_deflate_plan_destroy.restype = None
    return L


lib = _load()


def device_info() -> str:
    return lib.zsc_hip_device_info().decode()


# ---- sizing helpers ------------------------------------------------------------

def compress_get_min_work_buf_size(window_bits: int = DEF_WBITS, mem_level: int = DEF_MEM_LEVEL):
    out = C.c_uint32()
    rc = lib.zsc_compress_get_min_work_buf_size2(window_bits, mem_level, C.byref(out))
    return rc, out.value


def compress_get_max_output_size2(source_len: int, max_block_len: int, level: int,
                                  window_bits: int = DEF_WBITS, mem_level: int = DEF_MEM_LEVEL):
    out = C.c_uint32()
    rc = lib.zsc_compress_get_max_output_size2(source_len, max_block_len, level, window_bits,
                                               mem_level, C.byref(out))
    return rc, out.value


def compress_get_max_output_size(source_len: int, max_block_len: int, level: int):
    return compress_get_max_output_size2(source_len, max_block_len, level)


def uncompress_get_min_work_buf_size(window_bits: int = DEF_WBITS):
    out = C.c_uint32()
    rc = lib.zsc_uncompress_get_min_work_buf_size2(window_bits, C.byref(out))
    return rc, out.value


# ---- one-shot calls (reference zsc_pub.h:201-411) ---------------------------------

class GzHeader(C.Structure):
    """gz_header, reference include/zsc/zlib_types_pub.h:281-296"""
    _fields_ = [("text", C.c_int32), ("time", C.c_uint32), ("xflags", C.c_int32), ("os", C.c_int32),
                ("extra", C.c_void_p), ("extra_len", C.c_uint32), ("extra_max", C.c_uint32),
                ("name", C.c_void_p), ("name_max", C.c_uint32),
                ("comment", C.c_void_p), ("comm_max", C.c_uint32),
                ("hcrc", C.c_int32), ("done", C.c_int32)]


def gz_header_for_writing(text=0, time=0, os=3, extra: Optional[bytes] = None, name: Optional[bytes] = None,
                          comment: Optional[bytes] = None, hcrc=0):
    """A gz_header for zsc_compress_gzip*; returns (struct, buffers to keep alive)."""
    h = GzHeader()
    h.text, h.time, h.os, h.hcrc = text, time, os, hcrc
    keep = []
    if extra is not None:
        b = C.create_string_buffer(extra, max(len(extra), 1))
        keep.append(b)
        h.extra, h.extra_len = C.addressof(b), len(extra)
    if name is not None:
        b = C.create_string_buffer(name + b"\0")
        keep.append(b)
        h.name = C.addressof(b)
    if comment is not None:
        b = C.create_string_buffer(comment + b"\0")
        keep.append(b)
        h.comment = C.addressof(b)
    return h, keep


def gz_header_for_reading(extra_max=0, name_max=0, comm_max=0):
    """A gz_header for zsc_uncompress_gzip*; returns (struct, (extra, name, comment) buffers)."""
    h = GzHeader()
    bufs = []
    for cap, ptr, mx in ((extra_max, "extra", "extra_max"), (name_max, "name", "name_max"),
                         (comm_max, "comment", "comm_max")):
        b = C.create_string_buffer(max(cap, 1)) if cap else None
        bufs.append(b)
        if b is not None:
            setattr(h, ptr, C.addressof(b))
            setattr(h, mx, cap)
    return h, tuple(bufs)


def gz_header_fields(h: GzHeader, bufs) -> dict:
    """What a reader finds in the struct afterwards (buffers cut at their capacity)."""
    extra, name, comment = bufs
    return {"text": h.text, "time": h.time, "xflags": h.xflags, "os": h.os, "hcrc": h.hcrc, "done": h.done,
            "extra_len": h.extra_len,
            "extra": None if not h.extra or extra is None else extra.raw[:min(h.extra_len, h.extra_max)].hex(),
            "name": None if not h.name or name is None else name.raw.split(b"\0")[0].hex(),
            "comment": None if not h.comment or comment is None else comment.raw.split(b"\0")[0].hex()}


def compress2(source: bytes, max_block_len: Optional[int] = None, level: int = 6,
              window_bits: int = DEF_WBITS, mem_level: int = DEF_MEM_LEVEL,
              strategy: int = Z_DEFAULT_STRATEGY, dest_len: Optional[int] = None,
              work_len: Optional[int] = None, gz_header: Optional[GzHeader] = None) -> Tuple[int, bytes]:
    """zsc_compress2 / zsc_compress_gzip2 (reference zsc_pub.h:258,290).  Returns (ZlibReturn, stream bytes)."""
    n = len(source)
    mbl = max(n, 1) if max_block_len is None else max_block_len
    if dest_len is None:
        rc, dest_len = compress_get_max_output_size2(n, mbl, level, window_bits, mem_level)
        if rc != Z_OK:
            dest_len = n + (n >> 3) + 128
        if gz_header is not None:
            dest_len += 70000 * 3  # extra + name + comment at their largest
    if work_len is None:
        rc, work_len = compress_get_min_work_buf_size(window_bits, mem_level)
        if rc != Z_OK:
            work_len = 400000
    dst = C.create_string_buffer(max(dest_len, 1))
    work = C.create_string_buffer(max(work_len, 1))
    dl = C.c_uint32(dest_len)
    rc = lib.zsc_compress_gzip2(dst, C.byref(dl), source, n, mbl, work, work_len, level,
                                window_bits, mem_level, strategy,
                                None if gz_header is None else C.byref(gz_header))
    return rc, dst.raw[:dl.value]


def compress(source: bytes, max_block_len: Optional[int] = None, level: int = 6, **kw):
    """zsc_compress (reference zsc_pub.h:201): zlib wrapper, default window and memory."""
    return compress2(source, max_block_len, level, DEF_WBITS, DEF_MEM_LEVEL, Z_DEFAULT_STRATEGY, **kw)


def compress_gzip(source: bytes, max_block_len: Optional[int] = None, level: int = 6, **kw):
    """zsc_compress_gzip (reference zsc_pub.h:227) with gz_header == NULL."""
    return compress2(source, max_block_len, level, DEF_WBITS + GZIP_CODE, DEF_MEM_LEVEL,
                     Z_DEFAULT_STRATEGY, **kw)


def uncompress2(source: bytes, dest_len: int, window_bits: int = DEF_WBITS,
                work_len: Optional[int] = None, gz_header: Optional[GzHeader] = None) -> Tuple[int, bytes, int]:
    """zsc_uncompress2 (reference zsc_pub.h:385).  Returns (ZlibReturn, bytes, consumed)."""
    if work_len is None:
        rc, work_len = uncompress_get_min_work_buf_size(window_bits)
        if rc != Z_OK:
            work_len = 40000
    dst = C.create_string_buffer(max(dest_len, 1))
    work = C.create_string_buffer(max(work_len, 1))
    dl, sl = C.c_uint32(dest_len), C.c_uint32(len(source))
    rc = lib.zsc_uncompress_gzip2(dst, C.byref(dl), source, C.byref(sl), work, work_len,
                                  window_bits, None if gz_header is None else C.byref(gz_header))
    return rc, dst.raw[:dl.value], sl.value


def uncompress(source: bytes, dest_len: int, **kw):
    return uncompress2(source, dest_len, DEF_WBITS, **kw)


def uncompress_gzip(source: bytes, dest_len: int, **kw):
    return uncompress2(source, dest_len, DEF_WBITS + GZIP_CODE, **kw)


# ---- batches ---------------------------------------------------------------------

def compress_batch(sources: Sequence[bytes], level: int = 6, window_bits: int = DEF_WBITS,
                   mem_level: int = DEF_MEM_LEVEL, strategy: int = Z_DEFAULT_STRATEGY,
                   dest_caps: Optional[Sequence[int]] = None) -> Tuple[int, List[bytes], List[int]]:
    """zsc_hip_compress_batch: every item behaves like one zsc_compress2 call."""
    count = len(sources)
    if dest_caps is None:
        dest_caps = [compress_get_max_output_size2(len(s), max(len(s), 1), level, window_bits,
                                                   mem_level)[1] for s in sources]
    srcs = (C.c_char_p * count)(*sources)
    slen = (C.c_uint32 * count)(*[len(s) for s in sources])
    bufs = [C.create_string_buffer(max(c, 1)) for c in dest_caps]
    dsts = (C.c_void_p * count)(*[C.addressof(b) for b in bufs])
    dlen = (C.c_uint32 * count)(*dest_caps)
    stat = (C.c_int32 * count)()
    rc = lib.zsc_hip_compress_batch(count, srcs, slen, dsts, dlen, stat, level, window_bits,
                                    mem_level, strategy)
    outs = [bufs[i].raw[:dlen[i]] for i in range(count)] if rc == Z_OK else []
    return rc, outs, list(stat)


def compress_sections_batch(sources: Sequence[bytes], max_block_lens: Sequence[int], level: int = 6,
                            window_bits: int = DEF_WBITS, mem_level: int = DEF_MEM_LEVEL,
                            strategy: int = Z_DEFAULT_STRATEGY,
                            dest_caps: Optional[Sequence[int]] = None) -> Tuple[int, List[bytes], List[int]]:
    """zsc_hip_compress_sections_batch: item i behaves like zsc_compress2 with
    max_block_lens[i] < len(sources[i]) at levels 1-9 (sections, flush markers, output slices)."""
    count = len(sources)
    if dest_caps is None:
        dest_caps = [compress_get_max_output_size2(len(s), m, level, window_bits, mem_level)[1]
                     for s, m in zip(sources, max_block_lens)]
    srcs = (C.c_char_p * count)(*sources)
    slen = (C.c_uint32 * count)(*[len(s) for s in sources])
    mbls = (C.c_uint32 * count)(*max_block_lens)
    bufs = [C.create_string_buffer(max(c, 1)) for c in dest_caps]
    dsts = (C.c_void_p * count)(*[C.addressof(b) for b in bufs])
    dlen = (C.c_uint32 * count)(*dest_caps)
    stat = (C.c_int32 * count)()
    rc = lib.zsc_hip_compress_sections_batch(count, srcs, slen, mbls, dsts, dlen, stat, level,
                                             window_bits, mem_level, strategy, 0)
    outs = [bufs[i].raw[:dlen[i]] for i in range(count)] if rc == Z_OK else []
    return rc, outs, list(stat)


def compress_sections_device(d_input: int, in_offsets: Sequence[int], source_lens: Sequence[int],
                             max_block_lens: Sequence[int], d_output: int, out_offsets: Sequence[int],
                             out_caps: Sequence[int], level: int = 6, window_bits: int = DEF_WBITS,
                             mem_level: int = DEF_MEM_LEVEL,
                             strategy: int = Z_DEFAULT_STRATEGY) -> Tuple[int, List[int], List[int]]:
    """zsc_hip_compress_sections_device: streams and results stay in device memory
    (d_input / d_output are device pointers).  Returns (rc, stream lengths, statuses)."""
    count = len(source_lens)
    ioff = (C.c_uint64 * count)(*in_offsets)
    slen = (C.c_uint32 * count)(*source_lens)
    mbls = (C.c_uint32 * count)(*max_block_lens)
    ooff = (C.c_uint64 * count)(*out_offsets)
    caps = (C.c_uint32 * count)(*out_caps)
    dlen = (C.c_uint32 * count)()
    stat = (C.c_int32 * count)()
    rc = lib.zsc_hip_compress_sections_device(count, d_input, ioff, slen, mbls, d_output, ooff, caps,
                                              dlen, stat, level, window_bits, mem_level, strategy)
    return rc, list(dlen), list(stat)


def uncompress_batch(sources: Sequence[bytes], dest_caps: Sequence[int],
                     window_bits: int = DEF_WBITS) -> Tuple[int, List[bytes], List[int], List[int]]:
    """zsc_hip_uncompress_batch: every item behaves like one zsc_uncompress2 call.
    Returns (rc, outputs, consumed, statuses)."""
    count = len(sources)
    srcs = (C.c_char_p * count)(*sources)
    slen = (C.c_uint32 * count)(*[len(s) for s in sources])
    bufs = [C.create_string_buffer(max(c, 1)) for c in dest_caps]
    dsts = (C.c_void_p * count)(*[C.addressof(b) for b in bufs])
    dlen = (C.c_uint32 * count)(*dest_caps)
    stat = (C.c_int32 * count)()
    rc = lib.zsc_hip_uncompress_batch(count, srcs, slen, dsts, dlen, stat, window_bits)
    outs = [bufs[i].raw[:dlen[i]] for i in range(count)] if rc == Z_OK else []
    return rc, outs, list(slen), list(stat)


class InflatePlan:
    """Device-resident inflate batch (see include/zsc_hip.h)."""

    def __init__(self, source_lens: Sequence[int], dest_caps: Sequence[int],
                 window_bits: int = DEF_WBITS, decode_order: Sequence[int] | None = None):
        self.count = n = len(source_lens)
        so, do, sb, db = [], [], 0, 0
        for sl, dc in zip(source_lens, dest_caps):
            so.append(sb)
            do.append(db)
            sb += (sl + 64 + 15) & ~15
            db += (dc + 64 + 15) & ~15
        self.src_offsets, self.dst_offsets = so, do
        self.src_bytes, self.dst_bytes = sb + 64, db + 64
        self._h = C.c_void_p()
        order = None if decode_order is None else (C.c_uint32 * n)(*decode_order)
        rc = lib.zsc_hip_inflate_plan_create_ordered(C.byref(self._h), n, (C.c_uint32 * n)(*source_lens),
                                                     (C.c_uint64 * n)(*so), (C.c_uint32 * n)(*dest_caps),
                                                     (C.c_uint64 * n)(*do), window_bits, order)
        if rc != Z_OK:
            raise RuntimeError(f"zsc_hip_inflate_plan_create failed: {rc}")

    def run(self, d_src: int, d_dst: int, stream: int = 0) -> None:
        rc = lib.zsc_hip_inflate_plan_run(self._h, C.c_void_p(d_src), C.c_void_p(d_dst),
                                          C.c_void_p(stream))
        if rc != Z_OK:
            raise RuntimeError(f"zsc_hip_inflate_plan_run failed: {rc}")

    def results(self):
        n = self.count
        lens, used, stat, ms = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_int32 * n)(), C.c_float()
        rc = lib.zsc_hip_inflate_plan_results(self._h, lens, used, stat, C.byref(ms))
        if rc != Z_OK:
            raise RuntimeError(f"zsc_hip_inflate_plan_results failed: {rc}")
        return list(lens), list(used), list(stat), ms.value

    def close(self) -> None:
        if self._h:
            lib.zsc_hip_inflate_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeflatePlan:
    """A device-resident batch: fixed buffer lengths, inputs/outputs stay in HBM.

    ``layout`` gives the byte offsets of every buffer inside one input and one output
    allocation; ``run`` takes raw device pointers (e.g. ``tensor.data_ptr()``), so the
    binding itself needs neither torch nor numpy.
    """

    def __init__(self, source_lens: Sequence[int], level: int = 6, window_bits: int = DEF_WBITS,
                 mem_level: int = DEF_MEM_LEVEL, strategy: int = Z_DEFAULT_STRATEGY):
        self.count = n = len(source_lens)
        self.source_lens = list(source_lens)
        lens = (C.c_uint32 * n)(*source_lens)
        self._in_off = (C.c_uint64 * n)()
        self._out_off = (C.c_uint64 * n)()
        self._caps = (C.c_uint32 * n)()
        ib, ob = C.c_uint64(), C.c_uint64()
        rc = lib.zsc_hip_deflate_plan_layout(n, lens, level, window_bits, mem_level, self._in_off,
                                             self._out_off, self._caps, C.byref(ib), C.byref(ob))
        if rc != Z_OK:
            raise ValueError(f"zsc_hip_deflate_plan_layout failed: {rc}")
        self.in_bytes, self.out_bytes = ib.value, ob.value
        self.in_offsets = list(self._in_off)
        self.out_offsets = list(self._out_off)
        self.out_caps = list(self._caps)
        self._h = C.c_void_p()
        rc = lib.zsc_hip_deflate_plan_create(C.byref(self._h), n, lens, self._in_off, self._out_off,
                                             self._caps, level, window_bits, mem_level, strategy)
        if rc != Z_OK:
            raise RuntimeError(f"zsc_hip_deflate_plan_create failed: {rc}")

    @property
    def scratch_bytes(self) -> int:
        return lib.zsc_hip_deflate_plan_scratch_bytes(self._h)

    @property
    def sub_batches(self) -> int:
        return lib.zsc_hip_deflate_plan_sub_batches(self._h)

    def profile(self, enable: bool = True) -> None:
        lib.zsc_hip_deflate_plan_profile(self._h, 1 if enable else 0)

    def run(self, d_input: int, d_output: int, stream: int = 0) -> None:
        rc = lib.zsc_hip_deflate_plan_run(self._h, C.c_void_p(d_input), C.c_void_p(d_output),
                                          C.c_void_p(stream))
        if rc != Z_OK:
            raise RuntimeError(f"zsc_hip_deflate_plan_run failed: {rc}")

    def results(self) -> Tuple[List[int], List[int]]:
        lens = (C.c_uint32 * self.count)()
        stat = (C.c_int32 * self.count)()
        rc = lib.zsc_hip_deflate_plan_results(self._h, lens, stat)
        if rc != Z_OK:
            raise RuntimeError(f"zsc_hip_deflate_plan_results failed: {rc}")
        return list(lens), list(stat)

    def kernel_times_ms(self) -> dict:
        t = (C.c_float * NKERNELS)()
        rc = lib.zsc_hip_deflate_plan_times(self._h, t)
        if rc != Z_OK:
            raise RuntimeError("profiling was not enabled before the run")
        return dict(zip(KERNEL_NAMES, list(t)))

    def close(self) -> None:
        if self._h:
            lib.zsc_hip_deflate_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
