/*
 * zsc_api.c -- the reference's own one-shot API (include/zsc/zsc_pub.h) on top of
 * the MI355X runtime.  Plain C host code, as in the reference.
 *
 * Every function keeps the reference's name, argument meaning, validation order
 * and return codes (reference src/zsc_compress.c, src/zsc_uncompr.c, and the
 * parameter checks of deflateInit2_ / deflateWorkSize2 / deflateBoundNoStream /
 * inflateWorkSize2 that those wrappers run first).  The codec work itself is
 * handed to the HIP kernels through zsc_hip_compress_batch / _uncompress_batch
 * with a batch of one; there is no CPU codec in this library.
 */
#include "zsc/zsc_pub.h"
#include "zsc/zsc_conf_private.h"
#include "zsc_hip.h"

/* sizeof(deflate_state) / sizeof(inflate_state) of the reference on an LP64 host
 * (reference include/zsc/deflate.h:119-290, include/zsc/inflate.h:106-149; both are
 * below the Z_*_STATE_SIZE ceilings of zlib_types_pub.h:94,112).  The minimum work
 * sizes the reference demands derive from them; tests/test_api_host.py checks the
 * results against the compiled reference. */
#define ZSC_DEFLATE_STATE_BYTES 5920u
#define ZSC_INFLATE_STATE_BYTES 7152u

/* ---- sizing helpers -------------------------------------------------------- */

/* reference deflateWorkSize2, src/deflate.c:857-902 */
ZlibReturn zsc_compress_get_min_work_buf_size2(I32 window_bits, I32 mem_level, U32 *size_out)
{
    ZSC_ASSERT(size_out != Z_NULL);
    *size_out = U32_MAX;
    if (window_bits < 0) {
        window_bits = -window_bits;
    } else if (window_bits > 15) {
        window_bits -= 16;
    }
    if (window_bits == 8) {
        window_bits = 9;
    }
    if (mem_level < 1 || mem_level > MAX_MEM_LEVEL || window_bits < 8 || window_bits > 15) {
        ZSC_WARN2("In zsc_compress_get_min_work_buf_size2(), bad mem_level (%d) or "
                  "window_bits (%d).", mem_level, window_bits);
        return Z_STREAM_ERROR;
    }
    U32 w = 1u << window_bits;
    *size_out = ZSC_DEFLATE_STATE_BYTES + w * 2u         /* window */
                + w * 2u * (U32)sizeof(U16)              /* prev, reserved twice */
                + (1u << (mem_level + 7)) * (U32)sizeof(U16) /* head */
                + (1u << (mem_level + 6)) * 4u;          /* pending_buf */
    return Z_OK;
}

ZlibReturn zsc_compress_get_min_work_buf_size(U32 *size_out)
{
    return zsc_compress_get_min_work_buf_size2(DEF_WBITS, DEF_MEM_LEVEL, size_out);
}

/* reference deflateBoundNoStream, src/deflate.c:761-849 */
static ZlibReturn bound_no_stream(U32 source_len, I32 level, I32 window_bits, I32 mem_level,
                                  gz_header *head, U32 *size_out)
{
    ZSC_ASSERT(size_out != Z_NULL);
    *size_out = U32_MAX;
    I32 wrap = 1;
    if (window_bits < 0) {
        wrap = 0;
        window_bits = -window_bits;
    } else if (window_bits > 15) {
        wrap = 2;
        window_bits -= 16;
    }
    if (mem_level < 1 || mem_level > MAX_MEM_LEVEL || window_bits < 8 || window_bits > 15 ||
        (window_bits == 8 && wrap != 1)) {
        return Z_STREAM_ERROR;
    }
    U32 wraplen = 0;
    if (wrap == 1) {
        wraplen = 6 + 4;
    } else if (wrap == 2) {
        wraplen = 18;
        if (head != Z_NULL) {
            if (head->extra != Z_NULL) {
                wraplen += 2 + head->extra_len;
            }
            const U8 *s = head->name;
            if (s != Z_NULL) {
                wraplen++;
                while (*s) {
                    s++;
                    wraplen++;
                }
            }
            s = head->comment;
            if (s != Z_NULL) {
                wraplen++;
                while (*s) {
                    s++;
                    wraplen++;
                }
            }
            if (head->hcrc) {
                wraplen += 2;
            }
        }
    }
    if (window_bits != 15 || mem_level != 8 || level == Z_NO_COMPRESSION) {
        *size_out = source_len + ((source_len + 7) >> 3) + ((source_len + 63) >> 6) + 5 + wraplen;
    } else {
        *size_out = source_len + (source_len >> 12) + (source_len >> 14) + (source_len >> 25) +
                    13 - 6 + wraplen;
    }
    return Z_OK;
}

/* reference src/zsc_compress.c:207-236 */
ZlibReturn zsc_compress_get_max_output_size_gzip2(U32 source_len, U32 max_block_len, I32 level,
                                                  I32 window_bits, I32 mem_level,
                                                  gz_header *gz_header, U32 *size_out)
{
    U32 first = U32_MAX;
    ZlibReturn err = bound_no_stream(source_len, level, window_bits, mem_level, gz_header, &first);
    if (err != Z_OK) {
        ZSC_WARN1("In zsc_compress_get_max_output_size_gzip2(), could not get deflate output "
                  "bound, error %d.", err);
        return err;
    }
    ZSC_ASSERT(max_block_len != 0);
    U32 sections = first / max_block_len + 1;
    return bound_no_stream(source_len + sections * 4u, level, window_bits, mem_level, gz_header,
                           size_out);
}

ZlibReturn zsc_compress_get_max_output_size2(U32 source_len, U32 max_block_len, I32 level,
                                             I32 window_bits, I32 mem_level, U32 *size_out)
{
    return zsc_compress_get_max_output_size_gzip2(source_len, max_block_len, level, window_bits,
                                                  mem_level, Z_NULL, size_out);
}

ZlibReturn zsc_compress_get_max_output_size_gzip(U32 source_len, U32 max_block_len, I32 level,
                                                 gz_header *gz_header, U32 *size_out)
{
    return zsc_compress_get_max_output_size_gzip2(source_len, max_block_len, level,
                                                  DEF_WBITS + GZIP_CODE, DEF_MEM_LEVEL, gz_header,
                                                  size_out);
}

ZlibReturn zsc_compress_get_max_output_size(U32 source_len, U32 max_block_len, I32 level,
                                            U32 *size_out)
{
    return zsc_compress_get_max_output_size2(source_len, max_block_len, level, DEF_WBITS,
                                             DEF_MEM_LEVEL, size_out);
}

/* reference inflateWorkSize2, src/inflate.c:249-276 */
ZlibReturn zsc_uncompress_get_min_work_buf_size2(I32 window_bits, U32 *size_out)
{
    ZSC_ASSERT(size_out != Z_NULL);
    if (window_bits < 0) {
        window_bits = -window_bits;
    } else if (window_bits < 48) {
        window_bits &= 15;
    }
    if (window_bits && (window_bits < 8 || window_bits > 15)) {
        ZSC_WARN1("Cannot determine working size for windowBits = %d", window_bits);
        return Z_STREAM_ERROR;
    }
    *size_out = ZSC_INFLATE_STATE_BYTES + (1u << window_bits);
    return Z_OK;
}

ZlibReturn zsc_uncompress_get_min_work_buf_size(U32 *size_out)
{
    return zsc_uncompress_get_min_work_buf_size2(DEF_WBITS, size_out);
}

/* ---- compression ------------------------------------------------------------ */

/* reference src/zsc_compress.c:50-160 */
ZlibReturn zsc_compress_gzip2(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                              U32 max_block_len, U8 *work, U32 work_len, I32 level,
                              I32 window_bits, I32 mem_level, ZlibStrategy strategy,
                              gz_header *gz_header)
{
    ZSC_ASSERT(source != Z_NULL);
    ZSC_ASSERT(dest != Z_NULL);
    ZSC_ASSERT(dest_len != Z_NULL);
    ZSC_ASSERT(work != Z_NULL);

    const U32 dest_cap = *dest_len;
    *dest_len = 0;

    /* :74-88 work buffer large enough? (state lives in HBM, the contract stays) */
    U32 need = U32_MAX;
    ZlibReturn err = zsc_compress_get_min_work_buf_size2(window_bits, mem_level, &need);
    if (err != Z_OK) {
        ZSC_WARN1("In zsc_compress_gzip2(), could not get min work buf size, error %d.", err);
        return err;
    }
    if (work_len < need) {
        ZSC_WARN2("In zsc_compress_gzip2(), working memory (%u B) was smaller than required "
                  "(%u B).", work_len, need);
        return Z_MEM_ERROR;
    }

    /* deflateInit2_ parameter checks, reference src/deflate.c:305-328 */
    I32 lvl = level == Z_DEFAULT_COMPRESSION ? 6 : level;
    I32 wrap = 1, wb = window_bits;
    if (wb < 0) {
        wrap = 0;
        wb = -wb;
    } else if (wb > 15) {
        wrap = 2;
        wb -= 16;
    }
    if (mem_level < 1 || mem_level > MAX_MEM_LEVEL || wb < 8 || wb > 15 || lvl < 0 || lvl > 9 ||
        (I32)strategy < 0 || (I32)strategy > (I32)Z_FIXED || (wb == 8 && wrap != 1)) {
        ZSC_WARN4("In zsc_compress_gzip2(), bad arguments. memLevel:%d windowBits:%d level:%d "
                  "strategy:%d", mem_level, wb, lvl, (I32)strategy);
        return Z_STREAM_ERROR;
    }
    /* deflateSetHeader, reference src/deflate.c:545-556 */
    if (gz_header != Z_NULL && wrap != 2) {
        ZSC_WARN("In zsc_compress_gzip2(), could not set deflate header, error -2.");
        return Z_STREAM_ERROR;
    }
    /* :109-117 */
    U32 bound = U32_MAX;
    err = zsc_compress_get_max_output_size_gzip2(source_len, max_block_len, level, window_bits,
                                                 mem_level, gz_header, &bound);
    if (err != Z_OK) {
        return err;
    }
    ZSC_ASSERT(max_block_len != 0);

    /* what the kernels cover today; everything else fails loudly, never on a CPU path */
    if (source_len > max_block_len) {
        ZSC_WARN("In zsc_compress_gzip2(), multi-section streams (source_len > max_block_len) "
                 "are not offloaded yet.");
        return Z_STREAM_ERROR;
    }
    if (gz_header != Z_NULL) {
        ZSC_WARN("In zsc_compress_gzip2(), caller-supplied gzip header fields are not offloaded "
                 "yet.");
        return Z_STREAM_ERROR;
    }

    const U8 *srcs[1] = {source};
    U8 *dsts[1] = {dest};
    U32 slen[1] = {source_len};
    U32 dlen[1] = {dest_cap};
    I32 stat[1] = {Z_STREAM_ERROR};
    err = zsc_hip_compress_batch(1, srcs, slen, dsts, dlen, stat, level, window_bits, mem_level,
                                 strategy);
    if (err != Z_OK) {
        return err;
    }
    *dest_len = dlen[0];
    if (stat[0] != Z_OK) {
        ZSC_WARN1("In zsc_compress_gzip2(), deflate ended with error code %d.", stat[0]);
        if (dest_cap < bound) {
            ZSC_WARN2("In zsc_compress_gzip2(), output buffer (%u bytes) was smaller than the "
                      "bound (%u bytes). Output may not have fit in the buffer.", dest_cap, bound);
        }
    }
    return (ZlibReturn)stat[0];
}

ZlibReturn zsc_compress2(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                         U32 max_block_len, U8 *work, U32 work_len, I32 level, I32 window_bits,
                         I32 mem_level, ZlibStrategy strategy)
{
    return zsc_compress_gzip2(dest, dest_len, source, source_len, max_block_len, work, work_len,
                              level, window_bits, mem_level, strategy, Z_NULL);
}

ZlibReturn zsc_compress_gzip(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                             U32 max_block_len, U8 *work, U32 work_len, I32 level,
                             gz_header *gz_header)
{
    return zsc_compress_gzip2(dest, dest_len, source, source_len, max_block_len, work, work_len,
                              level, DEF_WBITS + GZIP_CODE, DEF_MEM_LEVEL, Z_DEFAULT_STRATEGY,
                              gz_header);
}

ZlibReturn zsc_compress(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                        U32 max_block_len, U8 *work, U32 work_len, I32 level)
{
    return zsc_compress2(dest, dest_len, source, source_len, max_block_len, work, work_len, level,
                         DEF_WBITS, DEF_MEM_LEVEL, Z_DEFAULT_STRATEGY);
}

/* ---- decompression ---------------------------------------------------------- */

/* reference src/zsc_uncompr.c:44-154 */
ZlibReturn zsc_uncompress_gzip2(U8 *dest, U32 *dest_len, const U8 *source, U32 *source_len,
                                U8 *work, U32 work_len, I32 window_bits, gz_header *gz_head)
{
    ZSC_ASSERT(source != Z_NULL);
    ZSC_ASSERT(source_len != Z_NULL);
    ZSC_ASSERT(dest != Z_NULL);
    ZSC_ASSERT(dest_len != Z_NULL);
    ZSC_ASSERT(work != Z_NULL);

    const U32 dest_cap = *dest_len, src_avail = *source_len;
    *dest_len = 0;
    *source_len = 0;

    U32 need = U32_MAX;
    ZlibReturn err = zsc_uncompress_get_min_work_buf_size2(window_bits, &need);
    if (err != Z_OK) {
        ZSC_WARN1("In zsc_uncompress_gzip2(), could not get work buffer size, error %d.", err);
        return err;
    }
    if (work_len < need) {
        ZSC_WARN2("In zsc_uncompress_gzip2(), work buffer (%u B) is smaller than required (%u B).",
                  work_len, need);
        return Z_MEM_ERROR;
    }
    if (gz_head != Z_NULL) {
        /* inflateGetHeader, reference src/inflate.c (wrap & 2 required) */
        if (window_bits < 16) {
            ZSC_WARN("In zsc_uncompress_gzip2(), could not get header, error -2.");
            return Z_STREAM_ERROR;
        }
        ZSC_WARN("In zsc_uncompress_gzip2(), returning gzip header fields is not offloaded yet.");
        return Z_STREAM_ERROR;
    }
    const U8 *srcs[1] = {source};
    U8 *dsts[1] = {dest};
    U32 slen[1] = {src_avail};
    U32 dlen[1] = {dest_cap};
    I32 stat[1] = {Z_STREAM_ERROR};
    err = zsc_hip_uncompress_batch(1, srcs, slen, dsts, dlen, stat, window_bits);
    if (err != Z_OK) {
        return err;
    }
    *dest_len = dlen[0];
    *source_len = slen[0];
    if (stat[0] != Z_OK) {
        ZSC_WARN1("In zsc_uncompress_gzip2(), inflate loop failed with error %d.", stat[0]);
    }
    return (ZlibReturn)stat[0];
}

ZlibReturn zsc_uncompress2(U8 *dest, U32 *dest_len, const U8 *source, U32 *source_len, U8 *work,
                           U32 work_len, I32 window_bits)
{
    return zsc_uncompress_gzip2(dest, dest_len, source, source_len, work, work_len, window_bits,
                                Z_NULL);
}

ZlibReturn zsc_uncompress(U8 *dest, U32 *dest_len, const U8 *source, U32 *source_len, U8 *work,
                          U32 work_len)
{
    return zsc_uncompress2(dest, dest_len, source, source_len, work, work_len, DEF_WBITS);
}

ZlibReturn zsc_uncompress_gzip(U8 *dest, U32 *dest_len, const U8 *source, U32 *source_len,
                               U8 *work, U32 work_len, gz_header *gz_head)
{
    return zsc_uncompress_gzip2(dest, dest_len, source, source_len, work, work_len,
                                DEF_WBITS + GZIP_CODE, gz_head);
}
