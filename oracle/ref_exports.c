/*
 * ref_exports.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Tiny export shim compiled together with the reference sources (where they lie
 * under /root/reference/src) into oracle/_ref/libzsc_ref.so.  The reference's
 * checksum entry points are called adler32()/crc32() (include/zsc/zlib.h:1153,
 * :1190) and would collide with the system libz that CPython already has
 * loaded, so the version script hides them and this file re-exports them under
 * zref_* names.  The 16 zsc_* functions are exported as they are.
 */
#include "zsc/zsc_pub.h"
#include "zsc/zlib.h"

U32 zref_adler32(U32 adler, const U8 *buf, U32 len) { return adler32(adler, buf, len); }
U32 zref_crc32(U32 crc, const U8 *buf, U32 len) { return crc32(crc, buf, len); }

/* raw pieces, for oracle unit checks */
U32 zref_deflate_bound_nostream(U32 source_len, I32 level, I32 window_bits, I32 mem_level,
                                U32 *size_out)
{
    return (U32)deflateBoundNoStream(source_len, level, window_bits, mem_level, Z_NULL, size_out);
}
