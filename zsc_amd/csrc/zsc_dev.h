/*
 * zsc_dev.h -- data layout shared by the host runtime and the gfx950 kernels.
 *
 * Vocabulary follows the reference (window, hash chain, block, lit/len + dist
 * symbols); see DESIGN.md "data layout in HBM".
 *
 *   input    all buffers of a batch back to back, each starting 16-byte aligned
 *   tiles    every buffer is cut into tiles of TILE = 32 768 positions (= the
 *            reference's w_size, src/deflate.c:344); one hash-sort workgroup per
 *            tile, tiles of all buffers numbered consecutively
 *   sorted   per tile, its positions ordered by (hash, position):
 *            entry = pos_in_tile | hash << 16           (u32, TILE per tile)
 *   rank     per input position, its index inside its tile's sorted array (u16)
 *   dir      per tile, first sorted index of every hash bucket (u16, DIR_STRIDE)
 *   symbols  per buffer, the LZ77 symbol stream: dist << 16 | lc   (u32)
 *   blocks   per buffer, the block cuts the parser made (BlockRec)
 *   plans    per block, Huffman code tables + header bits (BlockPlan)
 */
#ifndef ZSC_DEV_H
#define ZSC_DEV_H

#include <stdint.h>

#define ZD_TILE 32768u         /* w_size for window_bits 15 */
#define ZD_TILE_MASK 32767u
#define ZD_MAX_DIST 32506u     /* w_size - MIN_LOOKAHEAD, include/zsc/deflate.h:308 */
#define ZD_MIN_LOOKAHEAD 262u
#define ZD_MAX_MATCH 258u
#define ZD_TOO_FAR 4096u
#define ZD_HASH_BITS 15u       /* mem_level 8, src/deflate.c:347 */
#define ZD_HASH_MASK 0x7fffu
#define ZD_HASH_SHIFT 5u
#define ZD_SYM_CAP 16383u      /* lit_bufsize - 1, include/zsc/deflate.h:343 */
#define ZD_DIR_STRIDE 32776u   /* 32769 entries, padded to a multiple of 8 */
#define ZD_ENTRY_NONE 0xffffffffu

#define ZD_RING 36864u         /* LDS window ring of the parser: 32 KiB history + one 4 KiB chunk */
#define ZD_CHUNK 4096u

#define ZD_HDR_BYTES 320u      /* room for one dynamic block header (HLIT..code lengths) */

/* per-level search parameters, reference src/deflate.c:146-158 */
typedef struct {
    uint16_t good, lazy, nice, chain;
    uint32_t slow; /* 1: lazy parse (levels 4-9), 0: greedy (1-3) */
    /* from window_bits / mem_level (reference src/deflate.c:343-362): */
    uint32_t wsize;    /* 1 << window_bits: the window slides by this much */
    uint32_t max_dist; /* wsize - MIN_LOOKAHEAD */
    uint32_t sym_cap;  /* lit_bufsize - 1 = (1 << (mem_level + 6)) - 1 symbols per block */
    uint32_t hbits;    /* hash_bits = mem_level + 7; the chains are built on the 15-bit hash of
                          mem_level 8, which only matters in one corner (lz_head_blocked) */
} ZdLevel;

/* one buffer of a batch */
typedef struct {
    uint64_t in_off;    /* byte offset of the buffer in the batch input */
    uint64_t out_off;   /* byte offset of its stream in the batch output (4-byte aligned) */
    uint64_t sym_off;   /* first symbol slot */
    uint64_t rank_off;  /* first entry of the buffer in the rank array */
    uint32_t in_len;
    uint32_t out_cap;
    uint32_t tile0;     /* index of its first tile */
    uint32_t ntiles;
    uint32_t blk0;      /* index of its first BlockRec / BlockPlan */
    uint32_t max_blocks;
    uint32_t level;     /* 1..9 */
    uint32_t wrap;      /* 0 raw, 1 zlib, 2 gzip */
    uint32_t strategy;  /* 0 default, 1 filtered, 2 huffman only, 3 rle, 4 fixed */
    uint32_t wbits;     /* 9..15, for the zlib header */
    uint32_t more;      /* 1: a run of sections that is not the end of its stream -- the last block
                           is not final and only cut if it holds symbols (Z_FULL_FLUSH,
                           reference src/deflate.c:2118-2120) */
    uint32_t sched_off; /* first ZdSched of this run */
    uint32_t sched_n;
    uint32_t n0;        /* with joints: the length of the first section */
    uint32_t seg_ok;    /* with joints: the segmented parser may take the run (sections.h sec_seg_ok) */
    uint32_t pad;
} ZdBuf;

/* zsc_compress with source_len > max_block_len hands deflate() the input in sections
 * (reference src/zsc_compress.c:121-138).  A run = the sections parsed with one history.  How
 * much of the run deflate() has been given so far grows at joints, which the host finds by
 * following the output slices (sections.h):
 *   kind 0  when the block that is cut at `pos` for being full has been flushed
 *   kind 1  when the input given so far is used up (at `pos`): the owed literal goes out, the
 *           block is cut if it holds anything, and the parse carries on with the history
 * A joint of kind 0 whose cut lies MIN_LOOKAHEAD or more before the old end cannot have been
 * noticed by the parse before it -- the end of the input only shows within MIN_LOOKAHEAD of it
 * (lookahead caps, fill_window calls) -- so the parsers "fold" it: they take the new length from
 * the start of the phase.  pos == ZD_JOINT_ANYWHERE marks one the host only guesses (sections.h):
 * folded too, and checked against what really happens afterwards. */
#define ZD_JOINT_ANYWHERE 0xffffffffu
typedef struct {
    uint32_t pos;   /* relative to the start of the run */
    uint32_t new_n; /* the run's length from then on */
    uint32_t kind;
    uint32_t pad;
} ZdSched;

/* one entry of the match table (match_table.h): what longest_match returns at a position for one
 * value of prev_length, after the TOO_FAR / Z_FILTERED rule -- match_length in bits 0-8 (2 = no
 * match, or none longer than prev_length), distance - 1 in bits 9-23, flags in bits 30-31 */
#define MT_INCOMPLETE 0x80000000u /* not known: the parser searches itself */
#define MT_RLOK 0x40000000u       /* (r2 only) the entry also answers for any longer prev_length: the same
                                     match if it is longer than prev_length, none otherwise */
#define MT_LEN(e) ((e)&0x1ffu)
#define MT_DIST(e) ((((e) >> 9) & 0x7fffu) + 1u)
#define MT_PACK(len, dist) ((uint32_t)(len) | (((uint32_t)(dist)-1u) << 9))
#define MT_NONE MT_PACK(2u, 1u)

/* what the parser reports per buffer */
typedef struct {
    uint32_t nsyms;
    uint32_t nblocks;
} ZdParseOut;

/* one deflate block as cut by the parser (reference FLUSH_BLOCK, src/deflate.c:1660-1674) */
typedef struct {
    uint32_t sym_begin; /* relative to the buffer's sym_off */
    uint32_t sym_count;
    uint32_t in_begin;
    uint32_t in_len;
    uint32_t stored_ok;
    uint32_t last;
    uint32_t cut;      /* 0: the block was full (lit_bufsize - 1 symbols), 1: the input ended */
    uint32_t wend;     /* end of the window (base + 2 * w_size) when the block was cut: fill_window had
                          read up to there, or to the end of the input given so far if that is less */
    uint32_t at;       /* position of the parse-loop iteration that cut the block (the last one whose
                          fill_window call, if any, the window end above reflects) */
} ZdBlockRec;

#define ZD_CUT_FULL 0u
#define ZD_CUT_END 1u

#define ZD_BT_STORED 0u
#define ZD_BT_STATIC 1u
#define ZD_BT_DYNAMIC 2u

/* Huffman plan of one block (reference _tr_flush_block, src/trees.c:874-941) */
typedef struct {
    uint32_t type;       /* ZD_BT_* */
    uint32_t body_bits;  /* static/dynamic: 3-bit header + trees + symbols + END_BLOCK */
    uint32_t hdr_bits;   /* dynamic: bits in hdr[] (HLIT, HDIST, HCLEN, lengths) */
    uint32_t bit_off;    /* filled by the layout pass: first bit of the block in the stream */
    uint16_t lcode[286]; /* bit-reversed codes, ready to emit LSB first */
    uint8_t llen[286];
    uint16_t dcode[30];
    uint8_t dlen[30];
    uint8_t hdr[ZD_HDR_BYTES];
} ZdBlockPlan;

/* per-buffer result */
typedef struct {
    uint32_t out_len; /* bytes of the complete stream */
    int32_t status;   /* ZlibReturn */
    uint32_t adler;   /* adler32 or crc32 of the input */
    uint32_t bits;    /* bits of block data (a run that is not the end of its stream stops inside a byte) */
} ZdResult;

#endif
