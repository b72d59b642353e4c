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

/* sizeof(deflate_state) / sizeof(inflate_state) of the reference, which its minimum work sizes
 * derive from (reference src/deflate.c:857-902, src/inflate.c:249-276).  They depend on the
 * ABI (pointer size, alignment, enum width), so they are COMPUTED: the two structs below
 * restate the reference's private state member for member as far as layout goes -- same
 * sequence of types (include/zsc/deflate.h:119-290, include/zsc/inflate.h:106-149,
 * inftrees.h:57-61), runs of equally sized scalars written as arrays -- and the compiler
 * does the rest.  On LP64 they come to 5 920 and 7 152 bytes, what the compiled reference
 * reports (tests/test_api_host.py); both stay below the Z_*_STATE_SIZE ceilings of
 * zlib_types_pub.h. */
typedef enum { ZSC_LAYOUT_ENUM_0 = 0 } zsc_layout_enum; /* an enum as wide as the reference's own */
typedef struct { U16 fc, dl; } zsc_layout_ct;            /* ct_data: two unions of U16 */
typedef struct { void *dyn_tree; I32 max_code; const void *stat_desc; } zsc_layout_tree_desc;
struct zsc_layout_deflate_state {
    void *strm; I32 status;
    U8 *pending_buf; U32 pending_buf_size;
    U8 *pending_out; U32 pending_and_wrap[2];
    void *gzhead; U32 gzindex; ZlibMethod method; ZlibFlush last_flush;
    U32 w_size_bits_mask[3];
    U8 *window; U32 window_size;
    U16 *prev; U16 *head;
    U32 hash_and_match_state[16]; /* ins_h .. max_lazy_match, level */
    ZlibStrategy strategy; U32 good_match; I32 nice_match;
    zsc_layout_ct dyn_ltree[2 * 286 + 1], dyn_dtree[2 * 30 + 1], bl_tree[2 * 19 + 1];
    zsc_layout_tree_desc l_desc, d_desc, bl_desc;
    U16 bl_count[15 + 1];
    I32 heap[2 * 286 + 1], heap_len, heap_max;
    U8 depth[2 * 286 + 1];
    U8 *l_buf; U32 lit_bufsize, last_lit;
    U16 *d_buf; U32 opt_static_matches_insert[4];
    U16 bi_buf; I32 bi_valid; U32 high_water;
};
typedef struct { U8 op, bits; U16 val; } zsc_layout_code;
struct zsc_layout_inflate_state {
    void *strm; zsc_layout_enum mode; I32 last_wrap_havedict_flags[4]; U32 dmax_check_total[3];
    void *head; U32 wbits_wsize_whave_wnext[4];
    U8 *window; U32 hold_bits_length_offset_extra[5];
    const zsc_layout_code *lencode, *distcode; U32 lenbits_distbits_ncode_nlen_ndist_have[6];
    zsc_layout_code *next;
    U16 lens[320], work[288];
    zsc_layout_code codes[852 + 592];
    I32 sane, back; U32 was;
};
#define ZSC_DEFLATE_STATE_BYTES ((U32)sizeof(struct zsc_layout_deflate_state))
#define ZSC_INFLATE_STATE_BYTES ((U32)sizeof(struct zsc_layout_inflate_state))
ZSC_COMPILE_ASSERT(sizeof(struct zsc_layout_deflate_state) <= Z_DEFLATE_STATE_SIZE, deflate_state_fits_its_ceiling);
ZSC_COMPILE_ASSERT(sizeof(struct zsc_layout_inflate_state) <= Z_INFLATE_STATE_SIZE, inflate_state_fits_its_ceiling);

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

/* ---- gzip header fields (host side: a few dozen bytes per call) ---------------- */

/* CRC-32 of the gzip member header only (FHCRC); the data CRC is computed on the GPU */
static U32 hdr_crc32(U32 crc, const U8 *buf, U32 len)
{
    crc = ~crc;
    for (U32 i = 0; i < len; i++) {
        crc ^= buf[i];
        for (I32 k = 0; k < 8; k++) {
            crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
        }
    }
    return ~crc;
}

/* The member header deflate() writes for a caller-supplied gz_header, reference
 * src/deflate.c:1091-1200.  Writes at most `cap` bytes to `out`, returns the full length. */
static U32 gz_header_write(const gz_header *h, I32 level, ZlibStrategy strategy, U8 *out, U32 cap)
{
    U32 n = 0, crc = 0;
#define GZ_PUT(byte)                          \
    do {                                      \
        const U8 _b = (U8)(byte);             \
        if (h->hcrc) {                        \
            crc = hdr_crc32(crc, &_b, 1);     \
        }                                     \
        if (n < cap) {                        \
            out[n] = _b;                      \
        }                                     \
        n++;                                  \
    } while (0)
    GZ_PUT(31);
    GZ_PUT(139);
    GZ_PUT(8);
    GZ_PUT((h->text ? 1 : 0) + (h->hcrc ? 2 : 0) + (h->extra == Z_NULL ? 0 : 4) +
           (h->name == Z_NULL ? 0 : 8) + (h->comment == Z_NULL ? 0 : 16));
    GZ_PUT(h->time & 0xff);
    GZ_PUT((h->time >> 8) & 0xff);
    GZ_PUT((h->time >> 16) & 0xff);
    GZ_PUT((h->time >> 24) & 0xff);
    GZ_PUT(level == 9 ? 2 : (((I32)strategy >= (I32)Z_HUFFMAN_ONLY || level < 2) ? 4 : 0));
    GZ_PUT(h->os & 0xff);
    if (h->extra != Z_NULL) {
        GZ_PUT(h->extra_len & 0xff);
        GZ_PUT((h->extra_len >> 8) & 0xff);
        for (U32 i = 0; i < (h->extra_len & 0xffffu); i++) {
            GZ_PUT(h->extra[i]);
        }
    }
    if (h->name != Z_NULL) {
        U32 i = 0;
        U8 c;
        do {
            c = h->name[i++];
            GZ_PUT(c);
        } while (c != 0);
    }
    if (h->comment != Z_NULL) {
        U32 i = 0;
        U8 c;
        do {
            c = h->comment[i++];
            GZ_PUT(c);
        } while (c != 0);
    }
    if (h->hcrc) {
        const U32 c16 = crc; /* over everything before this field (HCRC_UPDATE, :1062-1068) */
        if (n < cap) {
            out[n] = (U8)(c16 & 0xff);
        }
        n++;
        if (n < cap) {
            out[n] = (U8)((c16 >> 8) & 0xff);
        }
        n++;
    }
#undef GZ_PUT
    return n;
}

/* What inflate() stores through inflateGetHeader while it reads the member header,
 * reference src/inflate.c:740-954: fields are filled as far as the input reaches; `done`
 * becomes 1 only after a complete (and, with FHCRC, verified) header, -1 for a zlib stream. */
static void gz_header_read(gz_header *h, const U8 *src, U32 avail)
{
    h->done = 0; /* inflateGetHeader */
    U32 pos = 0;
    if (avail < 2) {
        return;
    }
    if (!(src[0] == 31 && src[1] == 139)) {
        h->done = -1; /* :748-751 */
        return;
    }
    pos = 2;
    if (avail - pos < 2) {
        return;
    }
    const U32 flags = (U32)src[pos] | ((U32)src[pos + 1] << 8);
    if ((flags & 0xff) != 8 || (flags & 0xe000)) {
        return;
    }
    h->text = (I32)((flags >> 8) & 1);
    pos += 2;
    if (avail - pos < 4) {
        return;
    }
    h->time = (U32)src[pos] | ((U32)src[pos + 1] << 8) | ((U32)src[pos + 2] << 16) |
              ((U32)src[pos + 3] << 24);
    pos += 4;
    if (avail - pos < 2) {
        return;
    }
    h->xflags = (I32)src[pos];
    h->os = (I32)src[pos + 1];
    pos += 2;
    if (flags & 0x0400) {
        if (avail - pos < 2) {
            return;
        }
        const U32 xlen = (U32)src[pos] | ((U32)src[pos + 1] << 8);
        h->extra_len = xlen;
        pos += 2;
        U32 copy = xlen < avail - pos ? xlen : avail - pos;
        if (copy != 0 && h->extra != Z_NULL) {
            const U32 keep = copy > h->extra_max ? h->extra_max : copy;
            for (U32 i = 0; i < keep; i++) {
                h->extra[i] = src[pos + i];
            }
        }
        pos += copy;
        if (copy < xlen) {
            return;
        }
    } else {
        h->extra = Z_NULL;
    }
    if (flags & 0x0800) {
        if (avail == pos) {
            return;
        }
        U32 stored = 0;
        U8 c;
        do {
            c = src[pos++];
            if (h->name != Z_NULL && stored < h->name_max) {
                h->name[stored++] = c;
            }
        } while (c != 0 && pos < avail);
        if (c != 0) {
            return;
        }
    } else {
        h->name = Z_NULL;
    }
    if (flags & 0x1000) {
        if (avail == pos) {
            return;
        }
        U32 stored = 0;
        U8 c;
        do {
            c = src[pos++];
            if (h->comment != Z_NULL && stored < h->comm_max) {
                h->comment[stored++] = c;
            }
        } while (c != 0 && pos < avail);
        if (c != 0) {
            return;
        }
    } else {
        h->comment = Z_NULL;
    }
    if (flags & 0x0200) {
        if (avail - pos < 2) {
            return;
        }
        const U32 got = (U32)src[pos] | ((U32)src[pos + 1] << 8);
        if (got != (hdr_crc32(0, src, pos) & 0xffffu)) {
            return; /* header crc mismatch: the stream is bad, `done` stays 0 */
        }
    }
    h->hcrc = (I32)((flags >> 9) & 1);
    h->done = 1;
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

    if (lvl == 0) {
        /* deflate_stored: the layout follows the output slices; zsc_hip_store_batch does it,
         * sections and flush markers included */
        const U32 hlen = gz_header != Z_NULL ? gz_header_write(gz_header, lvl, strategy, dest, 0) : 0;
        const U8 *srcs0[1] = {source};
        U8 *dsts0[1] = {dest};
        U32 slen0[1] = {source_len}, mbl0[1] = {max_block_len}, dlen0[1] = {dest_cap};
        I32 stat0[1] = {Z_STREAM_ERROR};
        err = zsc_hip_store_batch(1, srcs0, slen0, mbl0, dsts0, dlen0, stat0, window_bits, mem_level,
                                  hlen);
        if (err != Z_OK) {
            return err;
        }
        if (gz_header != Z_NULL) {
            (void)gz_header_write(gz_header, lvl, strategy, dest, dlen0[0] < hlen ? dlen0[0] : hlen);
        }
        *dest_len = dlen0[0];
        if (stat0[0] != Z_OK) {
            ZSC_WARN1("In zsc_compress_gzip2(), deflate loop ended with error code %d.", stat0[0]);
        }
        return (ZlibReturn)stat0[0];
    }

    if (source_len > max_block_len) {
        /* sections with Z_FULL_FLUSH between them, and the output in slices of max_block_len
         * (reference src/zsc_compress.c:121-138): zsc_hip_compress_sections_batch */
        const U32 hlen = gz_header != Z_NULL ? gz_header_write(gz_header, lvl, strategy, dest, 0) : 0;
        const U8 *srcs1[1] = {source};
        U8 *dsts1[1] = {dest};
        U32 slen1[1] = {source_len}, mbl1[1] = {max_block_len}, dlen1[1] = {dest_cap};
        I32 stat1[1] = {Z_STREAM_ERROR};
        err = zsc_hip_compress_sections_batch(1, srcs1, slen1, mbl1, dsts1, dlen1, stat1, level,
                                              window_bits, mem_level, strategy, hlen);
        if (err != Z_OK) {
            return err;
        }
        if (gz_header != Z_NULL) {
            (void)gz_header_write(gz_header, lvl, strategy, dest, dlen1[0] < hlen ? dlen1[0] : hlen);
        }
        *dest_len = dlen1[0];
        if (stat1[0] != Z_OK) {
            ZSC_WARN1("In zsc_compress_gzip2(), deflate loop ended with error code %d.", stat1[0]);
        }
        return (ZlibReturn)stat1[0];
    }
    /* A caller-supplied gzip header only changes the member header: the stream is produced
     * with the plain 10-byte header placed so that it ends where the caller's header ends,
     * then the caller's header is written over the front (src/deflate.c:1091-1200). */
    U32 shift = 0;
    if (gz_header != Z_NULL) {
        const U32 hlen = gz_header_write(gz_header, lvl, strategy, dest, 0);
        if (dest_cap < hlen) {
            /* not even the header fits: the part that does is delivered, :1085-1090 + wrapper loop */
            (void)gz_header_write(gz_header, lvl, strategy, dest, dest_cap);
            *dest_len = dest_cap;
            ZSC_WARN1("In zsc_compress_gzip2(), deflate ended with error code %d.", Z_BUF_ERROR);
            return Z_BUF_ERROR;
        }
        shift = hlen - 10u;
    }

    const U8 *srcs[1] = {source};
    U8 *dsts[1] = {dest + shift};
    U32 slen[1] = {source_len};
    U32 dlen[1] = {dest_cap - shift};
    I32 stat[1] = {Z_STREAM_ERROR};
    err = zsc_hip_compress_batch(1, srcs, slen, dsts, dlen, stat, level, window_bits, mem_level,
                                 strategy);
    if (err != Z_OK) {
        return err;
    }
    if (gz_header != Z_NULL) {
        (void)gz_header_write(gz_header, lvl, strategy, dest, shift + 10u);
    }
    *dest_len = dlen[0] + shift;
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
        gz_head->done = 0;
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
    if (gz_head != Z_NULL) {
        gz_header_read(gz_head, source, src_avail);
    }
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
