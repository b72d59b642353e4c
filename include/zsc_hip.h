/*
 * zsc_hip.h -- batched / device-resident entry points of libzsc_hip.so.
 *
 * Extension of the reference API (SURVEY.md 8b "extension we add"): one
 * zsc_compress() call cannot amortise a kernel launch, so the library also
 * accepts MANY independent buffers per call.  Every item has exactly the
 * semantics of one reference zsc_compress2()/zsc_uncompress2() call
 * (src/zsc_compress.c:50-160, src/zsc_uncompr.c:44-154) with
 * max_block_len >= source_len: the stream produced for an item is byte-identical
 * to the reference's, and its status is the ZlibReturn the reference returns.
 *
 * C ABI only: plain pointers and sizes, no C++/torch types.  The reference-side
 * binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef ZSC_HIP_H
#define ZSC_HIP_H

#include <stdint.h>

#include "zsc/zlib_types_pub.h"

#ifdef __cplusplus
extern "C" {
#endif

/* library / device -------------------------------------------------------- */

/* 0 when a gfx950 device is usable; otherwise a negative ZlibReturn and a message
 * through ZSC_WARN.  Called implicitly by every other entry point. */
I32 zsc_hip_init(I32 device_ordinal);

/* human readable "device name | arch | CUs", valid until the next call */
const char *zsc_hip_device_info(void);

/* Device memory of finished calls and destroyed plans is kept for the next ones (at most a
 * quarter of the device's memory, or ZSC_HIP_CACHE_MB from the environment), which saves the
 * one-shot entry points their hipMalloc/hipFree time -- unmapping the scratch arrays was two
 * thirds of a sections call; this gives it back.  ZSC_HIP_NO_CACHE=1 keeps nothing. */
void zsc_hip_release_cached_memory(void);

/* host-pointer batches ---------------------------------------------------- */

/* Compress `count` independent buffers (host memory).  Item i:
 *   sources[i], source_lens[i]          input
 *   dests[i], dest_lens[i]              in: capacity of dests[i]; out: bytes written
 *   statuses[i]                         ZlibReturn of the item (Z_OK, Z_BUF_ERROR ...)
 * level / window_bits / mem_level / strategy as zsc_compress2 (reference
 * include/zsc/zsc_pub.h:258).  Returns Z_OK when the batch ran (look at statuses
 * for the items), or the error that stopped the whole batch. */
ZlibReturn zsc_hip_compress_batch(U32 count, const U8 *const *sources, const U32 *source_lens,
                                  U8 *const *dests, U32 *dest_lens, I32 *statuses, I32 level,
                                  I32 window_bits, I32 mem_level, ZlibStrategy strategy);

/* Level 0 (reference deflate_stored, src/deflate.c:1679-1880): `count` buffers stored, item i
 * like zsc_compress2(level 0, max_block_lens[i]) -- the lengths of the stored blocks follow the
 * output slices of max_block_len the wrapper hands out, and max_block_len < source_len gives
 * the sections their flush markers (src/zsc_compress.c:121-138). */
ZlibReturn zsc_hip_store_batch(U32 count, const U8 *const *sources, const U32 *source_lens,
                               const U32 *max_block_lens, U8 *const *dests, U32 *dest_lens,
                               I32 *statuses, I32 window_bits, I32 mem_level, U32 gzip_header_len);
/* gzip_header_len: 0, or the length of a caller-supplied gzip member header that the caller
 * writes over the start of each stream afterwards (zsc_compress_gzip with a gz_header) */

/* Levels 1-9 with source_lens[i] > max_block_lens[i]: item i like zsc_compress2(max_block_lens[i])
 * (reference src/zsc_compress.c:121-138) -- the input goes to deflate() in sections of
 * max_block_len with Z_FULL_FLUSH, the output in slices of max_block_len.  All sections of all
 * items are parsed at once; where an output slice ran out at a place that lets the next section
 * in early (SURVEY finding 2) the run is parsed again with the history (zsc_amd/csrc/sections.h).
 * gzip_header_len as in zsc_hip_store_batch. */
ZlibReturn zsc_hip_compress_sections_batch(U32 count, const U8 *const *sources,
                                           const U32 *source_lens, const U32 *max_block_lens,
                                           U8 *const *dests, U32 *dest_lens, I32 *statuses,
                                           I32 level, I32 window_bits, I32 mem_level,
                                           ZlibStrategy strategy, U32 gzip_header_len);

/* The same with the streams in device memory: stream i is source_lens[i] bytes at d_input +
 * in_offsets[i] (16-byte aligned) and its compressed stream goes to d_output + out_offsets[i],
 * of which out_caps[i] bytes may be used (the capacity is part of the result: the wrapper hands
 * it out in slices of max_block_len).  Synchronous: returns when the streams are in place. */
ZlibReturn zsc_hip_compress_sections_device(U32 count, const void *d_input,
                                            const uint64_t *in_offsets, const U32 *source_lens,
                                            const U32 *max_block_lens, void *d_output,
                                            const uint64_t *out_offsets, const U32 *out_caps,
                                            U32 *dest_lens, I32 *statuses, I32 level,
                                            I32 window_bits, I32 mem_level, ZlibStrategy strategy);

/* Decompress `count` independent streams (host memory).  source_lens[i]: in bytes
 * available, out bytes consumed (reference zsc_uncompress2, zsc_pub.h:385). */
ZlibReturn zsc_hip_uncompress_batch(U32 count, const U8 *const *sources, U32 *source_lens,
                                    U8 *const *dests, U32 *dest_lens, I32 *statuses,
                                    I32 window_bits);

/* device-resident plans --------------------------------------------------- */

/* A plan fixes the shape of a batch (how many buffers, how long each one is, the
 * codec parameters), owns all scratch memory in HBM and can be run many times on
 * inputs that already live in device memory.  Layout of the device buffers:
 *   input   buffer i occupies [in_offsets[i], in_offsets[i] + source_lens[i]);
 *           offsets are multiples of 16; 64 readable bytes must follow the last buffer
 *   output  stream i is written at out_offsets[i] (multiple of 16), capacity
 *           out_caps[i] >= zsc_compress_get_max_output_size(source_lens[i], ...)
 * zsc_hip_deflate_plan_layout() fills offsets/capacities with the tightest legal
 * layout and returns the two buffer sizes to allocate. */
typedef struct zsc_hip_deflate_plan zsc_hip_deflate_plan;

ZlibReturn zsc_hip_deflate_plan_layout(U32 count, const U32 *source_lens, I32 level,
                                       I32 window_bits, I32 mem_level, uint64_t *in_offsets,
                                       uint64_t *out_offsets, U32 *out_caps,
                                       uint64_t *in_bytes, uint64_t *out_bytes);

ZlibReturn zsc_hip_deflate_plan_create(zsc_hip_deflate_plan **plan, U32 count,
                                       const U32 *source_lens, const uint64_t *in_offsets,
                                       const uint64_t *out_offsets, const U32 *out_caps,
                                       I32 level, I32 window_bits, I32 mem_level,
                                       ZlibStrategy strategy);

/* Enqueue one full pass (checksum, hash sort, parse, Huffman plan, layout, bit
 * packing) on `hip_stream` (a hipStream_t, or NULL for the default stream).
 * Asynchronous: results are read with zsc_hip_deflate_plan_results(). */
ZlibReturn zsc_hip_deflate_plan_run(zsc_hip_deflate_plan *plan, const void *d_input,
                                    void *d_output, void *hip_stream);

/* Wait for the last run and fetch per-buffer sizes and statuses (either may be NULL). */
ZlibReturn zsc_hip_deflate_plan_results(zsc_hip_deflate_plan *plan, U32 *dest_lens,
                                        I32 *statuses);

/* Per-kernel device time in milliseconds, measured with HIP events recorded on the
 * run's stream and averaged over every run since profiling was switched on:
 * index 0 checksum, 1 hash sort, 2 match table (levels 4-9), 3 parse (the segmented multi-wave
 * kernel at levels 4-9, the greedy kernel at levels 1-3), 4 parse of short buffers (wave-per-buffer
 * kernels, levels 4-9), 5 huffman plan, 6 layout, 7 bit emit, 8 whole pass.  A plan that had to be
 * cut into sub-batches (zsc_hip_deflate_plan_sub_batches) launches every kernel once per
 * sub-batch; the times are sums over them.  Read after zsc_hip_deflate_plan_results(). */
#define ZSC_HIP_NKERNELS 9
void zsc_hip_deflate_plan_profile(zsc_hip_deflate_plan *plan, I32 enable);
ZlibReturn zsc_hip_deflate_plan_times(zsc_hip_deflate_plan *plan, float *ms_out);

/* bytes of HBM scratch the plan holds */
uint64_t zsc_hip_deflate_plan_scratch_bytes(const zsc_hip_deflate_plan *plan);

/* number of sub-batches (kernel launch sets) one run of the plan issues */
U32 zsc_hip_deflate_plan_sub_batches(const zsc_hip_deflate_plan *plan);

void zsc_hip_deflate_plan_destroy(zsc_hip_deflate_plan *plan);

/* device-resident inflate batches ------------------------------------------ */

/* Stream i occupies [src_offsets[i], +source_lens[i]) of the device input (offsets
 * multiples of 16, 64 readable bytes after the last stream) and decodes into
 * [dst_offsets[i], +dest_caps[i]) of the device output (offsets multiples of 16).
 * Results per stream as zsc_uncompress2 reports them: status, bytes written,
 * bytes consumed.  kernel_ms (may be NULL): device time of the last run from HIP
 * events on the run's stream. */
typedef struct zsc_hip_inflate_plan zsc_hip_inflate_plan;

ZlibReturn zsc_hip_inflate_plan_create(zsc_hip_inflate_plan **plan, U32 count,
                                       const U32 *source_lens, const uint64_t *src_offsets,
                                       const U32 *dest_caps, const uint64_t *dst_offsets,
                                       I32 window_bits);
/* The same with the order in which the streams are handed to the wavefronts given by the caller:
 * decode_order is a permutation of 0 .. count-1 (Z_STREAM_ERROR if it is not), NULL = the library's
 * own order (longest output first).  Four streams share a wavefront, so a caller that knows which
 * streams resemble each other can keep them apart or together; bench.py uses it to keep the replicas
 * of one member of its replicated batch from sharing wavefronts. */
ZlibReturn zsc_hip_inflate_plan_create_ordered(zsc_hip_inflate_plan **plan, U32 count,
                                               const U32 *source_lens, const uint64_t *src_offsets,
                                               const U32 *dest_caps, const uint64_t *dst_offsets,
                                               I32 window_bits, const U32 *decode_order);
ZlibReturn zsc_hip_inflate_plan_run(zsc_hip_inflate_plan *plan, const void *d_src, void *d_dst,
                                    void *hip_stream);
ZlibReturn zsc_hip_inflate_plan_results(zsc_hip_inflate_plan *plan, U32 *dest_lens,
                                        U32 *consumed, I32 *statuses, float *kernel_ms);
void zsc_hip_inflate_plan_destroy(zsc_hip_inflate_plan *plan);

#ifdef __cplusplus
}
#endif
#endif
