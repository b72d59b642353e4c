/*
 * zsc_oracle.h -- CPU restatement of the zsc DEFLATE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in zsc_amd/ or libzsc_hip.so may include,
 * link or call this; it exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check the HIP path against the reference's
 * algorithm on machines where /root/reference does not exist.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py compares every function
 * below byte-for-byte with the reference itself (oracle/_ref/libzsc_ref.so,
 * compiled from /root/reference/src by oracle/Makefile) on seeded inputs, and
 * tests/golden/ holds vectors generated from that reference build.
 *
 * The restatement is organised the way the GPU pipeline is:
 *   stage P  zo_parse()      LZ77 greedy/lazy parse  -> symbol stream + block cuts
 *   stage H  zo_block_plan() Huffman trees + stored/static/dynamic choice
 *   stage E  zo_emit()       bit packing
 * so each HIP kernel has a CPU stage with the same inputs and outputs.
 */
#ifndef ZSC_ORACLE_H
#define ZSC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes: the ZlibReturn values of include/zsc/zlib_types_pub.h */
#define ZO_OK 0
#define ZO_STREAM_ERROR (-2)
#define ZO_DATA_ERROR (-3)
#define ZO_MEM_ERROR (-4)
#define ZO_BUF_ERROR (-5)

/* one LZ77 symbol: dist == 0 -> literal byte `lc`; else match of length lc+3 */
typedef struct {
    uint16_t dist;
    uint8_t lc;
    uint8_t pad;
} zo_symbol;

/* one deflate block as cut by the parser (reference FLUSH_BLOCK, src/deflate.c:1660-1674) */
typedef struct {
    uint32_t sym_begin; /* first symbol of the block in the symbol stream */
    uint32_t sym_count; /* symbols in the block (<= lit_bufsize-1) */
    uint32_t in_begin;  /* first input byte covered */
    uint32_t in_len;    /* input bytes covered (stored_len) */
    uint8_t stored_ok;  /* input still in the sliding window at flush time */
    uint8_t last;       /* BFINAL */
    uint8_t pad[2];
} zo_block;

/* checksums: reference src/adler32.c:56-131, src/crc32.c:502-593 */
uint32_t zo_adler32(uint32_t adler, const uint8_t *buf, uint32_t len);
uint32_t zo_crc32(uint32_t crc, const uint8_t *buf, uint32_t len);

/* sizing: reference src/deflate.c:761-902, src/zsc_compress.c:207-236, src/inflate.c:249-276.
 * state_size stands for sizeof(deflate_state)/sizeof(inflate_state) of the build. */
int zo_deflate_bound(uint32_t source_len, int level, int window_bits, int mem_level,
                     uint32_t *size_out);
int zo_compress_max_output(uint32_t source_len, uint32_t max_block_len, int level,
                           int window_bits, int mem_level, uint32_t *size_out);
int zo_compress_work_size(int window_bits, int mem_level, uint32_t state_size,
                          uint32_t *size_out);
int zo_uncompress_work_size(int window_bits, uint32_t state_size, uint32_t *size_out);

/* stage P: parse `n` input bytes at `level` (1..9).  Writes at most n symbols and
 * at most n/((1<<(mem_level+6))-1)+2 blocks.  Returns ZO_OK / ZO_STREAM_ERROR. */
int zo_parse(const uint8_t *in, uint32_t n, int level, int window_bits_abs, int mem_level,
             int strategy, zo_symbol *syms, uint32_t *nsyms, zo_block *blocks,
             uint32_t *nblocks);

/* whole call: semantics of reference zsc_compress_gzip2 (src/zsc_compress.c:50-160)
 * with gz_header == NULL and a work buffer of `work_len` bytes.
 * Supported: level 0..9 (0 = deflate_stored, whose block lengths follow the output slices the
 * wrapper hands out), all five strategies, window_bits 9..15 in every wrapper, mem_level
 * 1..9, and source_len > max_block_len: the wrapper's section / output-slice loop with its
 * full-flush markers and the cases where a marker is skipped (src/zsc_compress.c:121-138,
 * src/deflate.c:1211-1264).  *unsupported stays 0 for every valid call now (it used to tell
 * "oracle cannot" from "reference says error"). */
int zo_compress(uint8_t *dest, uint32_t *dest_len, const uint8_t *source, uint32_t source_len,
                uint32_t max_block_len, uint32_t work_len, int level, int window_bits,
                int mem_level, int strategy, int *unsupported);

/* whole call: semantics of reference zsc_uncompress_gzip2 (src/zsc_uncompr.c:44-154)
 * with gz_head == NULL, including what it does after a data error: inflateSync looks for
 * the next full-flush marker and decoding goes on from there (src/inflate.c:1523-1604);
 * the call then ends in ZO_DATA_ERROR or ZO_BUF_ERROR with everything salvaged in dest. */
int zo_uncompress(uint8_t *dest, uint32_t *dest_len, const uint8_t *source,
                  uint32_t *source_len, uint32_t work_len, int window_bits);

#ifdef __cplusplus
}
#endif
#endif
