/*
 * zsc_pub.h -- the zsc one-shot compress / uncompress API, served by MI355X.
 *
 * Drop-in boundary.  The 16 entry points below have the
 * names, argument order, in/out conventions and return codes of the reference
 * header include/zsc/zsc_pub.h (line of each reference prototype cited per
 * function); libzsc_hip.so exports them with C linkage so a caller built
 * against the reference links unchanged.  Differences a caller can observe:
 *   - the work buffer is validated for size (Z_MEM_ERROR when too small, as in
 *     the reference) but the codec state lives in HBM, not in `work`;
 *   - the LZ77 / Huffman / checksum work runs as HIP kernels on gfx950;
 *   - configurations not yet offloaded (see DESIGN.md "out of scope") return
 *     Z_STREAM_ERROR after a ZSC_WARN instead of silently running on the CPU.
 *
 * Conventions (reference src/zsc_compress.c:50-160, src/zsc_uncompr.c:44-154):
 *   *dest_len    in: capacity of dest        out: bytes written
 *   *source_len  (uncompress) in: bytes available   out: bytes consumed
 *   window_bits  9..15 zlib wrapper, -9..-15 raw deflate, +GZIP_CODE gzip
 *   NULL dest/dest_len/source/source_len/work -> ZSC_ASSERT, not an error code
 *   gz_header pointers may be NULL
 *   success is Z_OK (never Z_STREAM_END)
 */
#ifndef ZSC_PUB_H
#define ZSC_PUB_H

#include "zsc/zsc_conf_global_types.h"
#include "zsc/zlib_types_pub.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sizing helpers (pure host arithmetic) ------------------------------- */

/* reference zsc_pub.h:86 -- work bytes zsc_compress() demands (default params) */
ZlibReturn zsc_compress_get_min_work_buf_size(U32 *size_out);

/* reference zsc_pub.h:100 */
ZlibReturn zsc_compress_get_min_work_buf_size2(I32 window_bits, I32 mem_level,
                                               U32 *size_out);

/* reference zsc_pub.h:116 -- upper bound of *dest_len after zsc_compress() */
ZlibReturn zsc_compress_get_max_output_size(U32 source_len, U32 max_block_len,
                                            I32 level, U32 *size_out);

/* reference zsc_pub.h:133 */
ZlibReturn zsc_compress_get_max_output_size_gzip(U32 source_len, U32 max_block_len,
                                                 I32 level, gz_header *gz_header,
                                                 U32 *size_out);

/* reference zsc_pub.h:154 */
ZlibReturn zsc_compress_get_max_output_size2(U32 source_len, U32 max_block_len,
                                             I32 level, I32 window_bits,
                                             I32 mem_level, U32 *size_out);

/* reference zsc_pub.h:176 */
ZlibReturn zsc_compress_get_max_output_size_gzip2(U32 source_len, U32 max_block_len,
                                                  I32 level, I32 window_bits,
                                                  I32 mem_level, gz_header *gz_header,
                                                  U32 *size_out);

/* reference zsc_pub.h:304 */
ZlibReturn zsc_uncompress_get_min_work_buf_size(U32 *size_out);

/* reference zsc_pub.h:316 */
ZlibReturn zsc_uncompress_get_min_work_buf_size2(I32 window_bits, U32 *size_out);

/* ---- compression ---------------------------------------------------------- */

/* reference zsc_pub.h:201 -- zlib wrapper, window 15, mem level 8 */
ZlibReturn zsc_compress(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                        U32 max_block_len, U8 *work, U32 work_len, I32 level);

/* reference zsc_pub.h:227 -- gzip wrapper, optional header fields */
ZlibReturn zsc_compress_gzip(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                             U32 max_block_len, U8 *work, U32 work_len, I32 level,
                             gz_header *gz_header);

/* reference zsc_pub.h:258 */
ZlibReturn zsc_compress2(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                         U32 max_block_len, U8 *work, U32 work_len, I32 level,
                         I32 window_bits, I32 mem_level, ZlibStrategy strategy);

/* reference zsc_pub.h:290 -- the general form all others forward to */
ZlibReturn zsc_compress_gzip2(U8 *dest, U32 *dest_len, const U8 *source, U32 source_len,
                              U32 max_block_len, U8 *work, U32 work_len, I32 level,
                              I32 window_bits, I32 mem_level, ZlibStrategy strategy,
                              gz_header *gz_header);

/* ---- decompression -------------------------------------------------------- */

/* reference zsc_pub.h:340 */
ZlibReturn zsc_uncompress(U8 *dest, U32 *dest_len, const U8 *source, U32 *source_len,
                          U8 *work, U32 work_len);

/* reference zsc_pub.h:362 */
ZlibReturn zsc_uncompress_gzip(U8 *dest, U32 *dest_len, const U8 *source,
                               U32 *source_len, U8 *work, U32 work_len,
                               gz_header *gz_head);

/* reference zsc_pub.h:385 */
ZlibReturn zsc_uncompress2(U8 *dest, U32 *dest_len, const U8 *source, U32 *source_len,
                           U8 *work, U32 work_len, I32 window_bits);

/* reference zsc_pub.h:409 -- the general form all others forward to */
ZlibReturn zsc_uncompress_gzip2(U8 *dest, U32 *dest_len, const U8 *source,
                                U32 *source_len, U8 *work, U32 work_len,
                                I32 window_bits, gz_header *gz_head);

#ifdef __cplusplus
}
#endif

#endif /* ZSC_PUB_H */
