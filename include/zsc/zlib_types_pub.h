/*
 * zlib_types_pub.h -- public types of the zsc one-shot API (MI355X drop-in).
 *
 * ABI mirror of the reference header include/zsc/zlib_types_pub.h: the same
 * enumerator values (:141-232), the same z_stream / gz_header field order
 * (:254-296) and the same compile-time sizing macros (:67-121), so a caller
 * compiled against the reference header links against libzsc_hip.so unchanged.
 * Written from the layout, not copied: comments and grouping are ours.
 */
#ifndef ZLIB_TYPES_PUB_H
#define ZLIB_TYPES_PUB_H

#include "zsc/zsc_conf_global_types.h"

ZSC_COMPILE_ASSERT(sizeof(U8) == 1, zsc_u8_size);
ZSC_COMPILE_ASSERT(sizeof(U16) == 2, zsc_u16_size);
ZSC_COMPILE_ASSERT(sizeof(U32) == 4, zsc_u32_size);
ZSC_COMPILE_ASSERT(sizeof(I32) == 4, zsc_i32_size);

#define Z_NULL 0

/* ---- compile-time buffer sizing (reference :67-121) --------------------- */

/* worst-case compressed size, gzip wrapper, no name/comment/extra */
#define Z_DEFLATE_OUTPUT_BOUND(source_len) \
    ((source_len) + (((source_len) + 7) >> 3) + (((source_len) + 63) >> 6) + 5 + 18 + 2)

/* the same when the stream is cut into sections of at least min_max_block_len */
#define Z_DEFLATE_OUTPUT_BOUND_BLOCKS(source_len, min_max_block_len) \
    (Z_DEFLATE_OUTPUT_BOUND((source_len)) + \
     (Z_DEFLATE_OUTPUT_BOUND((source_len)) / (min_max_block_len) + 1) * 4)

#define Z_DEFLATE_STATE_SIZE 6400
#define Z_INFLATE_STATE_SIZE 7600

/* window 2*w, prev reserved at 2*w U16 (reference src/deflate.c:895), head, pending */
#define Z_COMPRESS_WORK_SIZE2(window_bits, mem_level) \
    (Z_DEFLATE_STATE_SIZE + \
     (1 << (window_bits)) * 2 * sizeof(U8) + \
     (1 << (window_bits)) * 2 * sizeof(U16) + \
     (1 << ((mem_level) + 7)) * sizeof(U16) + \
     (1 << ((mem_level) + 6)) * (sizeof(U16) + 2))

#define Z_UNCOMPRESS_WORK_SIZE2(window_bits) \
    (Z_INFLATE_STATE_SIZE + (1 << (window_bits)) * sizeof(U8))

/* ---- constants ----------------------------------------------------------- */

enum {
    MAX_MEM_LEVEL = 9,
    DEF_MEM_LEVEL = 8,
    MAX_WBITS = 15, /* 32 KiB LZ77 window */
    DEF_WBITS = MAX_WBITS
};

enum {
    GZIP_CODE = 0x10 /* window_bits + GZIP_CODE selects the gzip wrapper */
};

typedef enum {
    Z_NO_FLUSH = 0,
    Z_PARTIAL_FLUSH = 1,
    Z_SYNC_FLUSH = 2,
    Z_FULL_FLUSH = 3,
    Z_FINISH = 4,
    Z_BLOCK = 5,
    Z_TREES = 6
} ZlibFlush;

typedef enum {
    Z_OK = 0,
    Z_STREAM_END = 1,
    Z_NEED_DICT = 2,
    Z_ERRNO = -1,
    Z_STREAM_ERROR = -2, /* bad parameter / inconsistent stream state */
    Z_DATA_ERROR = -3,   /* corrupt compressed input */
    Z_MEM_ERROR = -4,    /* work buffer too small */
    Z_BUF_ERROR = -5,    /* destination too small / input truncated */
    Z_VERSION_ERROR = -6
} ZlibReturn;

enum {
    Z_NO_COMPRESSION = 0,
    Z_BEST_SPEED = 1,
    Z_BEST_COMPRESSION = 9,
    Z_DEFAULT_COMPRESSION = -1
};

typedef enum {
    Z_FILTERED = 1,
    Z_HUFFMAN_ONLY = 2,
    Z_RLE = 3,
    Z_FIXED = 4,
    Z_DEFAULT_STRATEGY = 0
} ZlibStrategy;

typedef enum {
    Z_BINARY = 0,
    Z_TEXT = 1,
    Z_ASCII = Z_TEXT,
    Z_UNKNOWN = 2
} ZlibDataType;

typedef enum {
    Z_DEFLATED = 8
} ZlibMethod;

/* ---- stream descriptor (kept for ABI parity; the one-shot API builds it
 * internally, reference src/zsc_compress.c:65-72) -------------------------- */

struct internal_state;

typedef struct z_stream_s {
    const U8 *next_in;
    U32 avail_in;
    U32 total_in;

    U8 *next_out;
    U32 avail_out;
    U32 total_out;

    U8 *next_work; /* caller-supplied work memory replaces zalloc/zfree */
    U32 avail_work;

    const U8 *msg;
    struct internal_state *state;

    ZlibDataType data_type;
    U32 adler;
    U32 reserved;
} z_stream;

/* gzip member header fields, RFC 1952 */
typedef struct gz_header_s {
    I32 text;
    U32 time;
    I32 xflags;
    I32 os;
    U8 *extra;
    U32 extra_len;
    U32 extra_max;
    U8 *name;
    U32 name_max;
    U8 *comment;
    U32 comm_max;
    I32 hcrc;
    I32 done;
} gz_header;

#endif /* ZLIB_TYPES_PUB_H */
