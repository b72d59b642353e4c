/*
 * zsc_hip_runtime.hip -- kernels and host runtime of libzsc_hip.so (gfx950 only).
 *
 * The __global__ functions are thin: each one finds its unit of work (tile,
 * buffer or block) and calls the wavefront code in the headers of this
 * directory.  The host half owns HBM scratch, builds the work descriptors and
 * enqueues the six kernels of a deflate pass on one HIP stream.
 *
 * Path and boundary: this file implements the batched extension declared in
 * include/zsc_hip.h; zsc_api.c puts the reference's own zsc_pub.h signatures on
 * top of it.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "bit_emit.h"
#include "checksum.h"
#include "hash_sort.h"
#include "huff_plan.h"
#include "inflate.h"
#include "lz_parse.h"
#include "lz_parse_seg.h"
#include "match_table.h"
#include "lz_parse_simple.h"
#include "sections.h"
#include "zsc_dev.h"

#include "zsc/zsc_conf_private.h"
#include "zsc_hip.h"

/* ------------------------------------------------------------------------ */
/* kernels                                                                  */
/* ------------------------------------------------------------------------ */

/* kernel 0: one wavefront per buffer */
__global__ __launch_bounds__(64) void k_checksum(const uint8_t *__restrict__ in,
                                                 const ZdBuf *__restrict__ bufs,
                                                 ZdResult *__restrict__ res, uint32_t nbuf)
{
    __shared__ CkLds lds;
    const uint32_t b = blockIdx.x;
    if (b >= nbuf)
        return;
    const ZdBuf buf = bufs[b];
    uint32_t v = 0;
    if (buf.wrap == 1)
        v = ck_adler32(in + buf.in_off, buf.in_len);
    else if (buf.wrap == 2)
        v = ck_crc32(in + buf.in_off, buf.in_len, &lds);
    if (threadIdx.x == 0)
        res[b].adler = v;
}

/* level 0: one piece of a stored stream -- a stored block (3-bit header in its own byte,
 * LEN, NLEN, the bytes), a flush marker, or the wrapper header / trailer; the host has laid
 * the pieces out (StoreSim), this only moves bytes */
typedef struct {
    uint64_t src_off; /* first input byte of a stored block (batch input) */
    uint64_t dst_off; /* first output byte of the piece (batch output) */
    uint32_t len;     /* stored bytes */
    uint32_t kind;    /* 0 stored block, 1 flush marker, 2 wrapper header, 3 trailer, 4 room for a caller's
                         gzip header; sections.h: 5 the marker's four bytes after a run, 6 plain bytes,
                         7 one byte AND arg, then len - 1 zero bytes (the end of a run) */
    uint32_t arg;     /* block: BFINAL; header: CMF<<8|FLG or the gzip XFL; trailer / header: owning buffer */
    uint32_t buf;     /* owning buffer (for the check value) */
} ZdStorePiece;

__global__ __launch_bounds__(256) void k_store(const uint8_t *__restrict__ in,
                                               uint8_t *__restrict__ out,
                                               const ZdStorePiece *__restrict__ pieces,
                                               const ZdBuf *__restrict__ bufs,
                                               const ZdResult *__restrict__ res, uint32_t npieces)
{
    if (blockIdx.x >= npieces)
        return;
    const ZdStorePiece pc = pieces[blockIdx.x];
    uint8_t *o = out + pc.dst_off;
    if (pc.kind == 0u) {
        /* reference _tr_stored_block, src/trees.c:838-849 */
        if (threadIdx.x == 0) {
            o[0] = (uint8_t)pc.arg;
            o[1] = (uint8_t)pc.len;
            o[2] = (uint8_t)(pc.len >> 8);
            o[3] = (uint8_t)~pc.len;
            o[4] = (uint8_t)(~pc.len >> 8);
        }
        const uint8_t *s = in + pc.src_off;
        for (uint32_t i = threadIdx.x; i < pc.len; i += blockDim.x)
            o[5u + i] = s[i];
    } else if (pc.kind == 6u) {
        const uint8_t *s = in + pc.src_off;
        for (uint32_t i = threadIdx.x; i < pc.len; i += blockDim.x)
            o[i] = s[i];
    } else if (threadIdx.x == 0) {
        if (pc.kind == 7u) {
            o[0] = in[pc.src_off] & (uint8_t)pc.arg;
            if (pc.len > 1u)
                o[1] = 0;
            return;
        }
        /* the small pieces; sections.h: only the first pc.len bytes when dest ends inside one */
        uint8_t t[10];
        uint32_t nt = 0;
        if (pc.kind == 5u) { /* LEN = 0, NLEN = ~0 of Z_FULL_FLUSH's empty stored block */
            t[0] = 0, t[1] = 0, t[2] = 0xff, t[3] = 0xff;
            nt = 4;
        } else if (pc.kind == 1u) { /* Z_FULL_FLUSH's empty stored block, src/deflate.c:1240-1243 */
            t[0] = 0, t[1] = 0, t[2] = 0, t[3] = 0xff, t[4] = 0xff;
            nt = 5;
        } else if (pc.kind == 2u) {
            if (bufs[pc.buf].wrap == 1u) { /* src/deflate.c:1031-1049 */
                t[0] = (uint8_t)(pc.arg >> 8), t[1] = (uint8_t)pc.arg;
                nt = 2;
            } else { /* :1068-1082 */
                t[0] = 31, t[1] = 139, t[2] = 8;
                t[3] = t[4] = t[5] = t[6] = t[7] = 0;
                t[8] = (uint8_t)pc.arg, t[9] = 3;
                nt = 10;
            }
        } else if (pc.kind == 3u) {
            const uint32_t c = res[pc.buf].adler, n = bufs[pc.buf].in_len;
            if (bufs[pc.buf].wrap == 1u) { /* :1282-1286 */
                t[0] = (uint8_t)(c >> 24), t[1] = (uint8_t)(c >> 16), t[2] = (uint8_t)(c >> 8), t[3] = (uint8_t)c;
                nt = 4;
            } else { /* :1272-1281 */
                for (uint32_t k = 0; k < 4; k++) {
                    t[k] = (uint8_t)(c >> (8 * k));
                    t[4 + k] = (uint8_t)(n >> (8 * k));
                }
                nt = 8;
            }
        }
        if (pc.len != 0u && pc.len < nt)
            nt = pc.len;
        for (uint32_t k = 0; k < nt; k++)
            o[k] = t[k];
    }
}

/* kernel 1: one workgroup (HS_WAVES wavefronts) per 32 KiB tile */
__global__ __launch_bounds__(HS_WAVES * 64) void k_hash_sort(
    const uint8_t *__restrict__ in, const ZdBuf *__restrict__ bufs,
    const uint32_t *__restrict__ tile_owner, uint32_t *__restrict__ sorted,
    uint32_t *__restrict__ tmp, uint16_t *__restrict__ rank, uint16_t *__restrict__ dir,
    uint32_t ntiles)
{
    __shared__ HsLds lds;
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles)
        return;
    const ZdBuf buf = bufs[tile_owner[tile]];
    const uint32_t t = tile - buf.tile0;
    HsTile job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.start = t * ZD_TILE;
    const uint32_t owners = buf.in_len >= 3 ? buf.in_len - 2 : 0;
    job.m = owners > job.start ? min(owners - job.start, ZD_TILE) : 0u;
    job.sorted = sorted + (uint64_t)tile * ZD_TILE;
    job.tmp = tmp + (uint64_t)tile * ZD_TILE;
    job.rank = rank + buf.rank_off;
    job.dir = dir + (uint64_t)tile * ZD_DIR_STRIDE;
    job.dir_prev = nullptr;
    job.hib = nullptr;
    job.cnt = nullptr;
    const int w = (int)(threadIdx.x >> 6);
    for (int phase = 0; phase < HS_PHASES; phase++) {
        hash_sort_phase(job, &lds, w, phase);
        __syncthreads();
    }
}

/* kernel 1b: one workgroup per tile: chain lengths, and the link into the previous tile */
__global__ __launch_bounds__(HS_WAVES * 64) void k_link_prev(
    const uint8_t *__restrict__ in, const ZdBuf *__restrict__ bufs,
    const uint32_t *__restrict__ tile_owner, const uint16_t *__restrict__ dir,
    const uint16_t *__restrict__ rank, uint16_t *__restrict__ hib, uint32_t *__restrict__ cnt,
    uint32_t ntiles)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles)
        return;
    const ZdBuf buf = bufs[tile_owner[tile]];
    const uint32_t t = tile - buf.tile0;
    HsTile job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.start = t * ZD_TILE;
    const uint32_t owners = buf.in_len >= 3 ? buf.in_len - 2 : 0;
    job.m = owners > job.start ? min(owners - job.start, ZD_TILE) : 0u;
    job.sorted = nullptr;
    job.tmp = nullptr;
    job.rank = const_cast<uint16_t *>(rank) + buf.rank_off;
    job.dir = const_cast<uint16_t *>(dir) + (uint64_t)tile * ZD_DIR_STRIDE;
    job.dir_prev = t ? dir + (uint64_t)(tile - 1) * ZD_DIR_STRIDE : nullptr;
    job.hib = hib + buf.rank_off;
    job.cnt = cnt + buf.rank_off;
    hs_link_prev(job, (int)(threadIdx.x >> 6));
}

/* kernel 1c: one workgroup per tile: longest_match for every position of the tile, for the two
 * values of prev_length the lazy parse can ask with (match_table.h) */
#ifdef MT_EU
#define MT_ATTR __attribute__((amdgpu_waves_per_eu(MT_EU, MT_EU)))
#else
#define MT_ATTR
#endif
__global__ __launch_bounds__(MT_WAVES * 64) MT_ATTR void k_match_table(
    const uint8_t *__restrict__ in, const ZdBuf *__restrict__ bufs,
    const uint32_t *__restrict__ tile_owner, const uint32_t *__restrict__ sorted,
    const uint16_t *__restrict__ rank, const uint16_t *__restrict__ hib,
    const uint32_t *__restrict__ cnt, uint32_t *__restrict__ r2,
    const ZdLevel cfg, uint32_t min_len, uint32_t cap, uint32_t ntiles)
{
    __shared__ MtLds lds;
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles)
        return;
    const ZdBuf buf = bufs[tile_owner[tile]];
    /* only what the segmented parser takes without joints reads the table */
    if (buf.in_len <= min_len || buf.sched_n != 0)
        return;
    MtJob job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.start = (tile - buf.tile0) * ZD_TILE;
    job.sorted = sorted + (uint64_t)buf.tile0 * ZD_TILE;
    job.rank = rank + buf.rank_off;
    job.hib = hib + buf.rank_off;
    job.cnt = cnt + buf.rank_off;
    job.r2 = r2 + buf.rank_off;
    job.cfg = cfg;
    job.cfg.wsize = ZD_TILE;
    job.cfg.max_dist = ZD_MAX_DIST;
    job.cfg.sym_cap = ZD_SYM_CAP;
    job.cfg.hbits = 15u;
    job.strategy = buf.strategy;
    job.cap = cap;
    const int w = (int)(threadIdx.x >> 6);
    mt_phase_load(job, &lds, w);
    __syncthreads();
    const uint32_t base0 = sg_base(job.cfg, job.start, job.n);
    mt_phase_search(job, &lds, w, base0);
}

/* kernel 2: one wavefront per buffer, longest buffers first.  L picks the LDS ring
 * size (and with it the number of waves a CU can hold); `first`/`nbuf` select the
 * slice of the length-sorted order this launch covers. */
template <class L>
__global__ __launch_bounds__(64) void k_parse(const uint8_t *__restrict__ in,
                                              const ZdBuf *__restrict__ bufs,
                                              const uint32_t *__restrict__ order,
                                              const uint32_t *__restrict__ sorted,
                                              const uint16_t *__restrict__ rank,
                                              const uint16_t *__restrict__ hib,
                                              uint32_t *__restrict__ syms,
                                              ZdBlockRec *__restrict__ recs,
                                              ZdParseOut *__restrict__ pout,
                                              const ZdSched *__restrict__ sched,
                                              const ZdLevel cfg, uint32_t first, uint32_t nbuf)
{
    __shared__ L lds;
    if (blockIdx.x >= nbuf)
        return;
    const uint32_t b = order[first + blockIdx.x];
    const ZdBuf buf = bufs[b];
    LzJob job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.ntot = buf.in_len;
    job.sorted = sorted + (uint64_t)buf.tile0 * ZD_TILE;
    job.rank = rank + buf.rank_off;
    job.hib = hib + buf.rank_off;
    job.cnt = nullptr;
    job.dir = nullptr;
    job.r2 = nullptr;
    job.stair_min = 0xffffffffu;
    job.syms = syms + buf.sym_off;
    job.blocks = recs + buf.blk0;
    job.out = pout + b;
    job.cfg = cfg;
    job.strategy = buf.strategy;
    job.more = buf.more;
    job.sched = sched + buf.sched_off;
    job.nsched = buf.sched_n;
    job.n0 = buf.n0;
    lz_parse_lazy<L>(job, &lds);
}

/* kernel 2 for long buffers at levels 4-9: SG_W wavefronts share one window and parse
 * SG_W segments of the same buffer at once (lz_parse_seg.h) */
template <bool GENERIC, bool TABLE, int LEVEL> /* GENERIC false: window_bits 15 / mem_level 8, their constants folded in;
                                                  TABLE: the plan has a match table (only with GENERIC false);
                                                  LEVEL 6: the default level's search parameters folded in too (0: any) */
__global__ __launch_bounds__(SG_W * 64, SG_MIN_WAVES) void k_parse_seg(const uint8_t *__restrict__ in,
                                                         const ZdBuf *__restrict__ bufs,
                                                         const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ sorted,
                                                         const uint16_t *__restrict__ rank,
                                                         const uint16_t *__restrict__ hib,
                                                         const uint32_t *__restrict__ cnt,
                                                         const uint16_t *__restrict__ dir,
                                                         const uint32_t *__restrict__ r2,
                                                         uint32_t *__restrict__ syms,
                                                         ZdBlockRec *__restrict__ recs,
                                                         ZdParseOut *__restrict__ pout,
                                                         uint32_t *__restrict__ seg_tok,
                                                         const ZdSched *__restrict__ sched,
                                                         const ZdLevel cfg, uint32_t stair_min,
                                                         uint32_t first, uint32_t nbuf)
{
    __shared__ SgLds lds;
    if (blockIdx.x >= nbuf)
        return;
    const uint32_t b = order[first + blockIdx.x];
    const ZdBuf buf = bufs[b];
    LzJob job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.ntot = buf.in_len;
    job.sorted = sorted + (uint64_t)buf.tile0 * ZD_TILE;
    job.rank = rank + buf.rank_off;
    job.hib = hib + buf.rank_off;
    job.cnt = cnt + buf.rank_off;
    /* given the directories, the parser works hib / cnt out itself and k_link_prev has not run (sg_link) */
    job.dir = dir ? dir + (uint64_t)buf.tile0 * ZD_DIR_STRIDE : nullptr;
    /* the match table covers plain buffers (match_table.h); a run with joints is searched as before */
    job.r2 = r2 && buf.sched_n == 0 ? r2 + buf.rank_off : nullptr;
    job.stair_min = stair_min;
    job.syms = syms + buf.sym_off;
    job.blocks = recs + buf.blk0;
    job.out = pout + b;
    job.cfg = cfg;
    if (!GENERIC) {
        job.cfg.wsize = ZD_TILE;
        job.cfg.max_dist = ZD_MAX_DIST;
        job.cfg.sym_cap = ZD_SYM_CAP;
        job.cfg.hbits = 15u;
    }
    if (LEVEL == 6) { /* reference src/deflate.c:155: the fewer wave-uniform values the parser keeps, the fewer it spills */
        job.cfg.good = 8;
        job.cfg.lazy = 16;
        job.cfg.nice = 128;
        job.cfg.chain = 128;
        job.cfg.slow = 1;
    }
    job.strategy = LEVEL == 6 ? 0u : buf.strategy; /* (LEVEL 6 is only launched for plans without Z_FILTERED, the one strategy the lazy parse looks at) */
    job.more = buf.more;
    job.sched = sched + buf.sched_off;
    job.nsched = buf.sched_n;
    job.n0 = buf.n0;
    SgScratch scr;
    /* one area per workgroup: the segments' tokens, then their token indices (one pointer to keep) */
    scr.tok = seg_tok + (uint64_t)blockIdx.x * SG_SCRATCH_WORDS;
    scr.sidx = (uint16_t *)(scr.tok + SG_NS * SG_TOKCAP);
    const int w = (int)(threadIdx.x >> 6);
    sg_init(&lds, w);
    __syncthreads();
    /* every loop is bounded so that a logic error can never hang the device: a phase needs
     * n/SG_SPAN + 1 super-steps, a run has at most one phase per joint and one more */
    const uint32_t max_steps = buf.in_len / SG_SPAN + 2;
    bool stuck = false;
    uint32_t si = 0, nph = buf.sched_n ? buf.n0 : buf.in_len;
    for (uint32_t phase = 0; phase <= buf.sched_n && !stuck; phase++) {
        nph = sg_phase_end(job, nph, &si); /* joints of kind 0 only move the end (lz_parse_seg.h) */
        const bool goes_on = si < buf.sched_n;
        job.n = nph;
        job.more = goes_on ? 1u : buf.more;
        uint32_t steps = 0;
        while (!lds.finished) {
            if (++steps > max_steps) {
                stuck = true;
                break;
            }
            sg_phase_begin(job, &lds, w);
            __syncthreads();
            /* a redo round follows whenever a parser gave up before it met a successor's
             * tokens; each one moves the resolver at least one segment on */
            uint32_t rounds = 0;
            do {
                if (++rounds > SG_NS + 1) {
                    stuck = true;
                    break;
                }
                if (TABLE && job.r2 != nullptr)
                    sg_phase_parse<true>(job, &lds, scr, w);
                else
                    sg_phase_parse<false>(job, &lds, scr, w);
                __syncthreads();
                sg_phase_resolve(job, &lds, scr, w);
                __syncthreads();
            } while (lds.redo);
            if (stuck)
                break;
        }
        if (!goes_on)
            break;
        /* every wave must have seen lds.finished before wave 0 clears it for the next phase */
        __syncthreads();
        nph = sched[buf.sched_off + si].new_n;
        si++;
        sg_next_phase(&lds, w, job.n);
        __syncthreads();
    }
    if (stuck && threadIdx.x == 0) {
        job.out->nsyms = 0;
        job.out->nblocks = 0xffffffffu; /* reported as Z_STREAM_ERROR by the layout kernel */
    }
}

/* kernel 2 for Z_HUFFMAN_ONLY and Z_RLE: no chains, no window (lz_parse_simple.h) */
__global__ __launch_bounds__(64) void k_parse_simple(const uint8_t *__restrict__ in,
                                                     const ZdBuf *__restrict__ bufs,
                                                     uint32_t *__restrict__ syms,
                                                     ZdBlockRec *__restrict__ recs,
                                                     ZdParseOut *__restrict__ pout,
                                                     const ZdSched *__restrict__ sched,
                                                     const ZdLevel cfg, uint32_t nbuf)
{
    __shared__ SpLds lds;
    const uint32_t b = blockIdx.x;
    if (b >= nbuf)
        return;
    const ZdBuf buf = bufs[b];
    LzJob job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.ntot = buf.in_len;
    job.sorted = nullptr;
    job.rank = nullptr;
    job.hib = nullptr;
    job.cnt = nullptr;
    job.dir = nullptr;
    job.r2 = nullptr;
    job.stair_min = 0xffffffffu;
    job.syms = syms + buf.sym_off;
    job.blocks = recs + buf.blk0;
    job.out = pout + b;
    job.cfg = cfg;
    job.strategy = buf.strategy;
    job.more = buf.more;
    job.sched = sched + buf.sched_off;
    job.nsched = buf.sched_n;
    job.n0 = buf.n0;
    if (buf.sched_n)
        lz_parse_simple_joints(job, &lds);
    else if (buf.strategy == (uint32_t)Z_HUFFMAN_ONLY)
        lz_parse_huff(job, &lds);
    else
        lz_parse_rle(job, &lds);
}

/* kernel 2 for levels 1-3 (greedy parse) */
template <class L> /* LzLdsFastG: the window read from the input; LzLdsFast: an LDS ring (four waves per CU) */
__global__ __launch_bounds__(64) void k_parse_fast(const uint8_t *__restrict__ in,
                                                   const ZdBuf *__restrict__ bufs,
                                                   const uint32_t *__restrict__ order,
                                                   const uint32_t *__restrict__ sorted,
                                                   const uint16_t *__restrict__ rank,
                                                   const uint16_t *__restrict__ hib,
                                                   uint32_t *__restrict__ syms,
                                                   ZdBlockRec *__restrict__ recs,
                                                   ZdParseOut *__restrict__ pout,
                                                   const ZdSched *__restrict__ sched,
                                                   const ZdLevel cfg, uint32_t nbuf)
{
    __shared__ L lds;
    if (blockIdx.x >= nbuf)
        return;
    const uint32_t b = order[blockIdx.x];
    const ZdBuf buf = bufs[b];
    LzJob job;
    job.in = in + buf.in_off;
    job.n = buf.in_len;
    job.ntot = buf.in_len;
    job.sorted = sorted + (uint64_t)buf.tile0 * ZD_TILE;
    job.rank = rank + buf.rank_off;
    job.hib = hib + buf.rank_off;
    job.cnt = nullptr;
    job.dir = nullptr;
    job.r2 = nullptr;
    job.stair_min = 0xffffffffu;
    job.syms = syms + buf.sym_off;
    job.blocks = recs + buf.blk0;
    job.out = pout + b;
    job.cfg = cfg;
    job.strategy = buf.strategy;
    job.more = buf.more;
    job.sched = sched + buf.sched_off;
    job.nsched = buf.sched_n;
    job.n0 = buf.n0;
    lz_parse_greedy<L>(job, &lds);
}

/* kernel 3: one wavefront per (possible) block */
__global__ __launch_bounds__(64) void k_huff_plan(const ZdBuf *__restrict__ bufs,
                                                  const uint32_t *__restrict__ blk_owner,
                                                  const uint32_t *__restrict__ syms,
                                                  const ZdBlockRec *__restrict__ recs,
                                                  const ZdParseOut *__restrict__ pout,
                                                  ZdBlockPlan *__restrict__ plans, uint32_t nslots)
{
    __shared__ HpLds lds;
    const uint32_t slot = blockIdx.x;
    if (slot >= nslots)
        return;
    const uint32_t b = blk_owner[slot];
    const ZdBuf buf = bufs[b];
    if (slot - buf.blk0 >= pout[b].nblocks || pout[b].nblocks > buf.max_blocks)
        return;
    const ZdBlockRec *rec = &recs[slot];
    huff_plan_block(syms + buf.sym_off + rec->sym_begin, rec, buf.strategy, &plans[slot], &lds);
}

/* kernel 4a: one thread per buffer */
__global__ __launch_bounds__(64) void k_layout(const ZdBuf *__restrict__ bufs,
                                               const ZdParseOut *__restrict__ pout,
                                               const ZdBlockRec *__restrict__ recs,
                                               ZdBlockPlan *__restrict__ plans,
                                               ZdResult *__restrict__ res,
                                               uint8_t *__restrict__ out, uint32_t nbuf)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuf)
        return;
    layout_buffer(&bufs[b], &pout[b], recs + bufs[b].blk0, plans + bufs[b].blk0, &res[b],
                  out + bufs[b].out_off);
}

/* sections.h wants to know where every block ends: the bit_off column of the plans, packed */
__global__ __launch_bounds__(256) void k_bit_offs(const ZdBlockPlan *__restrict__ plans,
                                                  uint32_t *__restrict__ out, uint32_t nslots)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots)
        out[i] = plans[i].bit_off;
}

/* kernel 4b: one wavefront per block */
__global__ __launch_bounds__(64) void k_emit(const uint8_t *__restrict__ in,
                                             const ZdBuf *__restrict__ bufs,
                                             const uint32_t *__restrict__ blk_owner,
                                             const uint32_t *__restrict__ syms,
                                             const ZdBlockRec *__restrict__ recs,
                                             const ZdParseOut *__restrict__ pout,
                                             const ZdBlockPlan *__restrict__ plans,
                                             uint8_t *__restrict__ out, uint32_t nslots)
{
    __shared__ BeLds lds;
    const uint32_t slot = blockIdx.x;
    if (slot >= nslots)
        return;
    const uint32_t b = blk_owner[slot];
    const ZdBuf buf = bufs[b];
    if (slot - buf.blk0 >= pout[b].nblocks || pout[b].nblocks > buf.max_blocks)
        return;
    const ZdBlockRec *rec = &recs[slot];
    emit_block(in + buf.in_off, syms + buf.sym_off + rec->sym_begin, rec, &plans[slot],
               (uint32_t *)(out + buf.out_off), &lds);
}

/* one stream of an inflate batch */
typedef struct {
    uint64_t src_off, dst_off;
    uint32_t src_len, dst_cap;
} ZdInfItem;

/* kernel 5: one group of INF_GROUP lanes per compressed stream, 64 / INF_GROUP streams per wavefront */
#define INF_PER_WAVE (64 / INF_GROUP)
#ifndef INF_WAVES_EU
#define INF_WAVES_EU 5 /* 96 VGPRs, with 8 KiB of LDS per wave (512-byte stages): 20 waves per CU */
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(INF_WAVES_EU, INF_WAVES_EU))) void k_inflate(const uint8_t *__restrict__ src,
                                                uint8_t *__restrict__ dst,
                                                const ZdInfItem *__restrict__ items,
                                                const uint32_t *__restrict__ order,
                                                InfResult *__restrict__ res,
                                                InfResume *__restrict__ resume,
                                                uint32_t *__restrict__ pending, int32_t window_bits,
                                                uint32_t count)
{
    __shared__ InfLds lds_all[INF_PER_WAVE];
    __shared__ uint32_t crc_table[1][256];
    const uint32_t grp = (threadIdx.x & 63u) / INF_GROUP;
    InfLds *lds = &lds_all[grp];
    if ((threadIdx.x & (INF_GROUP - 1u)) == 0)
        lds->cktab = crc_table;
    /* every group takes streams from one queue (pending[1], longest output first) until it is
     * empty: streams of one length still differ a lot in decoding time (a table-like member has
     * a third of a text member's symbols), and with a fixed four streams per wavefront the
     * groups that finish early idle until the slowest one is done */
    for (;;) {
        uint32_t slot = 0;
        if ((threadIdx.x & (INF_GROUP - 1u)) == 0)
            slot = atomicAdd(pending + 1, 1u);
        slot = (uint32_t)__shfl((int)slot, (int)(threadIdx.x & (64u - INF_GROUP)));
        if (slot >= count)
            break;
        const uint32_t i = order[slot];
        if (resume[i].state == 2u)
            continue; /* finished in an earlier launch (only streams that resynchronise come back) */
        const ZdInfItem it = items[i];
        InfJob job;
        job.src = src + it.src_off;
        job.n = it.src_len;
        job.dst = dst + it.dst_off;
        job.cap = it.dst_cap;
        job.window_bits = window_bits;
        if (inflate_stream(job, lds, &res[i], &resume[i])) {
            if ((threadIdx.x & (INF_GROUP - 1u)) == 0)
                atomicAdd(pending, 1u); /* the host launches once more for these */
        }
    }
}

/* ------------------------------------------------------------------------ */
/* host runtime                                                             */
/* ------------------------------------------------------------------------ */

namespace {

const ZdLevel kLevels[10] = {
    {0, 0, 0, 0, 0},         {4, 4, 8, 4, 0},       {4, 5, 16, 8, 0},     {4, 6, 32, 32, 0},
    {4, 4, 16, 16, 1},       {8, 16, 32, 32, 1},    {8, 16, 128, 128, 1}, {8, 32, 128, 256, 1},
    {32, 128, 258, 1024, 1}, {32, 258, 258, 4096, 1}};

std::once_flag g_init_once;
int g_init_status = Z_STREAM_ERROR;
int g_device = -1; /* the device this process's plans, scratch and kernels live on */
int g_cus = 256;    /* its compute units */
char g_device_info[256] = "uninitialised";

#define HIP_TRY(expr, fail)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ZSC_WARN2("zsc_hip: %s failed: %s", #expr, hipGetErrorString(_e));           \
            fail;                                                                        \
        }                                                                                \
    } while (0)

size_t g_dev_cache_cap = 4ull << 30; /* DevCache: bytes of freed device memory kept at most */

void do_init(int ordinal)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        ZSC_WARN("zsc_hip: no HIP device is visible; this library has no CPU path.");
        g_init_status = Z_STREAM_ERROR;
        return;
    }
    if (ordinal < 0) {
        /* one process per GPU: LOCAL_RANK names it; without it, the device the calling thread
         * works on already (never silently device 0 under somebody else's feet) */
        const char *lr = getenv("LOCAL_RANK");
        int cur = 0;
        if (lr)
            ordinal = atoi(lr) % count;
        else
            ordinal = hipGetDevice(&cur) == hipSuccess ? cur : 0;
    }
    if (hipSetDevice(ordinal) != hipSuccess) {
        ZSC_WARN1("zsc_hip: cannot select device %d.", ordinal);
        g_init_status = Z_STREAM_ERROR;
        return;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) {
        g_init_status = Z_STREAM_ERROR;
        return;
    }
    g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    snprintf(g_device_info, sizeof g_device_info, "%s | %s | %d CUs | %.1f GiB", prop.name,
             prop.gcnArchName, prop.multiProcessorCount,
             (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ZSC_WARN1("zsc_hip: built for gfx950 only, found %s.", prop.gcnArchName);
        g_init_status = Z_STREAM_ERROR;
        return;
    }
    g_dev_cache_cap = prop.totalGlobalMem / 4;
    if (const char *e = getenv("ZSC_HIP_CACHE_MB"))
        g_dev_cache_cap = (size_t)atoll(e) << 20;
    g_device = ordinal;
    g_init_status = Z_OK;
}

/* HIP's current device is a property of the calling THREAD.  Every entry point that touches
 * the device runs inside one of these: it selects the library's device for the call and puts
 * the thread's own choice back afterwards, so a caller (or another library) that works on a
 * different GPU is not disturbed, and a second thread does not launch on its default device
 * against memory that lives on ours. */
void ensure_init();
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    DeviceScope()
    {
        ensure_init();
        if (g_device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != g_device)
            changed = hipSetDevice(g_device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (changed)
            (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

/* Freed device blocks, kept for the next plan.  The one-shot entry points (zsc_compress ...)
 * build and drop a plan per call and the sections path one per round; hipMalloc / hipFree cost
 * a tenth of a millisecond each, hipFree waits for the device and takes ~30 ms per GB it
 * unmaps -- with ~16 B of scratch per input byte that was two thirds of a sections call.  At
 * most g_dev_cache_cap bytes (a quarter of the device's memory; ZSC_HIP_CACHE_MB) in kMaxBlocks blocks
 * are kept (ZSC_HIP_NO_CACHE: none); a block serves a request of at least half its size. */
struct DevCache {
    static constexpr size_t kMaxBlocks = 96;
    std::mutex mu;
    std::vector<std::pair<size_t, void *>> blocks;
    size_t held = 0;
    bool off = getenv("ZSC_HIP_NO_CACHE") != nullptr;

    void *take(size_t need, size_t *got)
    {
        std::lock_guard<std::mutex> lock(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i].first >= need && blocks[i].first <= 2 * need + (1u << 16) &&
                (best == blocks.size() || blocks[i].first < blocks[best].first))
                best = i;
        if (best == blocks.size())
            return nullptr;
        void *p = blocks[best].second;
        *got = blocks[best].first;
        held -= blocks[best].first;
        blocks[best] = blocks.back();
        blocks.pop_back();
        return p;
    }
    bool give(void *p, size_t bytes)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (off || blocks.size() >= kMaxBlocks || held + bytes > g_dev_cache_cap)
            return false;
        blocks.push_back(std::make_pair(bytes, p));
        held += bytes;
        return true;
    }
    void trim()
    {
        std::lock_guard<std::mutex> lock(mu);
        for (auto &b : blocks)
            (void)hipFree(b.second);
        blocks.clear();
        held = 0;
    }
};
DevCache g_dev_cache;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t need)
    {
        if (need <= bytes)
            return true;
        release();
        if (need == 0)
            need = 16;
        p = g_dev_cache.take(need, &bytes);
        if (p)
            return true;
        if (hipMalloc(&p, need) != hipSuccess) {
            (void)hipGetLastError(); /* (the error is sticky: it must not be charged to the next launch) */
            g_dev_cache.trim(); /* what is kept may be what is missing */
            if (hipMalloc(&p, need) != hipSuccess) {
                (void)hipGetLastError();
                p = nullptr;
                ZSC_WARN1("zsc_hip: hipMalloc of %zu bytes failed.", need);
                return false;
            }
        }
        bytes = need;
        return true;
    }
    /* the caller has waited for the work that used the block */
    void release()
    {
        if (p && !g_dev_cache.give(p, bytes))
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

/* the longest buffer one plan entry may be: deflateBound of it stays below 2^29 bytes = 2^32 bits */
#define ZSC_HIP_MAX_BUFFER 0x1ff00000u

/* a group of buffers that shares one set of scratch arrays */
struct SubBatch {
    uint32_t first = 0, count = 0; /* buffers [first, first+count) of the plan */
    uint32_t ntiles = 0, nslots = 0;
    uint64_t nsym_slots = 0;
    DevBuf d_bufs, d_tile_owner, d_blk_owner, d_order;
    /* the length-sorted order (longest first) splits into ring-size classes:
     * [0,c36) full ring, [c36,c16) <= 18 432 B, [c16,c8) <= 10 240 B, [c8,count) <= 6 144 B */
    uint32_t c36 = 0, c16 = 0, c8 = 0;
    uint32_t cseg = 0;  /* [0,cseg) of the length-sorted order go to the segmented parser */
};

} // namespace

struct zsc_hip_deflate_plan {
    uint32_t count = 0;
    int level = 6, wrap = 1, wbits = 15, mem_level = 8;
    uint32_t strategy = 0;
    std::vector<ZdBuf> bufs; /* in_off / out_off are absolute in the caller's buffers */
    std::vector<SubBatch> subs;
    /* scratch shared by all sub-batches (sized for the largest) */
    DevBuf d_sorted, d_tmp_syms, d_rank, d_hib, d_cnt, d_dir, d_recs, d_plans, d_pout;
    DevBuf d_seg_tok; /* per long buffer: token staging of the segmented parser (tokens, then token indices) */
    DevBuf d_r2; /* the match table (match_table.h): one entry per input position */
    bool use_table = false;
    uint32_t table_min = 0; /* buffers longer than this have a table (those the segmented parser takes) */
    uint32_t table_cap = MT_CAP;
    uint32_t stair_min = SG_STAIR_MIN; /* chains at least this long are searched as a staircase (lz_parse_seg.h) */
    DevBuf d_sched;               /* joints of runs of sections (sections.h) */
    bool use_seg = true;
    DevBuf d_res; /* one ZdResult per buffer of the whole plan */
    uint64_t rank_base_off = 0;
    bool profile = false;
    float times[ZSC_HIP_NKERNELS] = {0};
    std::vector<hipEvent_t> events; /* ZSC_HIP_NKERNELS per sub-batch per profiled run */
    size_t events_used = 0;
    uint32_t profiled_runs = 0;
    hipStream_t last_stream = nullptr;
    uint64_t scratch_bytes = 0;
};

namespace {
void ensure_init()
{
    int before = -1;
    (void)hipGetDevice(&before);
    std::call_once(g_init_once, do_init, -1);
    if (before >= 0 && getenv("LOCAL_RANK") == nullptr)
        (void)hipSetDevice(before); /* (do_init selects the device it adopts; a no-op here) */
}
} // namespace

extern "C" I32 zsc_hip_init(I32 device_ordinal)
{
    int before = -1;
    (void)hipGetDevice(&before);
    std::call_once(g_init_once, do_init, (int)device_ordinal);
    if (g_init_status == Z_OK && device_ordinal >= 0 && device_ordinal != g_device) {
        ZSC_WARN2("zsc_hip: already initialised on device %d, cannot move to device %d.", g_device,
                  (int)device_ordinal);
        return Z_STREAM_ERROR;
    }
    /* the calling thread keeps the device it had unless it asked for this one by number */
    if (device_ordinal < 0 && before >= 0 && getenv("LOCAL_RANK") == nullptr)
        (void)hipSetDevice(before);
    return g_init_status;
}

extern "C" void zsc_hip_release_cached_memory(void)
{
    DeviceScope scope;
    g_dev_cache.trim();
}

extern "C" const char *zsc_hip_device_info(void)
{
    (void)zsc_hip_init(-1);
    return g_device_info;
}

/* reference deflateBoundNoStream + zsc_compress_get_max_output_size2; defined in zsc_api.c */
extern "C" ZlibReturn zsc_compress_get_max_output_size2(U32, U32, I32, I32, I32, U32 *);

static bool offloadable(I32 level, I32 window_bits, I32 mem_level, ZlibStrategy strategy, int *wrap,
                        int *wbits)
{
    int wb = window_bits;
    *wrap = 1;
    if (wb < 0) {
        *wrap = 0;
        wb = -wb;
    } else if (wb > 15) {
        *wrap = 2;
        wb -= 16;
    }
    if (wb == 8 && *wrap == 1)
        wb = 9; /* reference src/deflate.c:329-331 */
    *wbits = wb;
    if (level == Z_DEFAULT_COMPRESSION)
        level = 6;
    return wb >= 9 && wb <= 15 && mem_level >= 1 && mem_level <= 9 && level >= 1 && level <= 9 &&
           (strategy == Z_DEFAULT_STRATEGY || strategy == Z_FILTERED || strategy == Z_FIXED ||
            strategy == Z_HUFFMAN_ONLY || strategy == Z_RLE);
}

extern "C" ZlibReturn zsc_hip_deflate_plan_layout(U32 count, const U32 *source_lens, I32 level,
                                                  I32 window_bits, I32 mem_level,
                                                  uint64_t *in_offsets, uint64_t *out_offsets,
                                                  U32 *out_caps, uint64_t *in_bytes,
                                                  uint64_t *out_bytes)
{
    ZSC_ASSERT(source_lens != Z_NULL);
    ZSC_ASSERT(in_offsets != Z_NULL);
    ZSC_ASSERT(out_offsets != Z_NULL);
    ZSC_ASSERT(out_caps != Z_NULL);
    uint64_t io = 0, oo = 0;
    for (U32 i = 0; i < count; i++) {
        U32 cap = 0;
        const U32 n = source_lens[i];
        ZlibReturn rc = zsc_compress_get_max_output_size2(n, n ? n : 1, level, window_bits,
                                                          mem_level, &cap);
        if (rc != Z_OK)
            return rc;
        in_offsets[i] = io;
        out_offsets[i] = oo;
        out_caps[i] = cap;
        io += ((uint64_t)n + 15u) & ~15ull;
        oo += ((uint64_t)cap + 16u + 15u) & ~15ull;
    }
    if (in_bytes)
        *in_bytes = io + 64;
    if (out_bytes)
        *out_bytes = oo + 64;
    return Z_OK;
}

namespace {
/* what a plan of runs of sections (sections.h) knows beyond the lengths */
struct PlanRuns {
    const uint32_t *more;      /* per buffer: ZdBuf.more */
    const uint32_t *n0;        /* per buffer: length of the first section */
    const uint32_t *sched_off; /* per buffer: first joint in sched[] */
    const uint32_t *sched_n;
    const uint32_t *seg_ok;    /* per buffer: sections.h sec_seg_ok */
    const ZdSched *sched;
    uint32_t nsched;
};
} // namespace

static ZlibReturn plan_create(zsc_hip_deflate_plan **plan_out, U32 count, const U32 *source_lens,
                              const uint64_t *in_offsets, const uint64_t *out_offsets,
                              const U32 *out_caps, I32 level, I32 window_bits, I32 mem_level,
                              ZlibStrategy strategy, const PlanRuns *runs);

extern "C" ZlibReturn zsc_hip_deflate_plan_create(zsc_hip_deflate_plan **plan_out, U32 count,
                                                  const U32 *source_lens,
                                                  const uint64_t *in_offsets,
                                                  const uint64_t *out_offsets, const U32 *out_caps,
                                                  I32 level, I32 window_bits, I32 mem_level,
                                                  ZlibStrategy strategy)
{
    return plan_create(plan_out, count, source_lens, in_offsets, out_offsets, out_caps, level,
                       window_bits, mem_level, strategy, nullptr);
}

static ZlibReturn plan_create(zsc_hip_deflate_plan **plan_out, U32 count, const U32 *source_lens,
                              const uint64_t *in_offsets, const uint64_t *out_offsets,
                              const U32 *out_caps, I32 level, I32 window_bits, I32 mem_level,
                              ZlibStrategy strategy, const PlanRuns *runs)
{
    DeviceScope scope;
    ZSC_ASSERT(plan_out != Z_NULL);
    *plan_out = nullptr;
    if (zsc_hip_init(-1) != Z_OK)
        return Z_STREAM_ERROR;
    int wrap = 1, wbits = 15;
    if (!offloadable(level, window_bits, mem_level, strategy, &wrap, &wbits)) {
        ZSC_WARN4("zsc_hip: level %d / window_bits %d / mem_level %d / strategy %d is not "
                  "offloaded to the GPU yet (DESIGN.md, out of scope).",
                  level, window_bits, mem_level, (int)strategy);
        return Z_STREAM_ERROR;
    }
    if (level == Z_DEFAULT_COMPRESSION)
        level = 6;
    auto *pl = new zsc_hip_deflate_plan();
    pl->count = count;
    pl->level = level;
    pl->wrap = wrap;
    pl->wbits = wbits;
    pl->mem_level = mem_level;
    pl->strategy = (uint32_t)strategy;
    pl->bufs.resize(count);

    /* input bytes per sub-batch; the scratch is ~14 B per input byte OF ONE SUB-BATCH.  Measured on
     * the x4096 batch (11.5 GB): 8 GiB sub-batches 3 641 MB/s with 165 GB of scratch, 4 GiB 3 631 MB/s
     * with 83 GB, 2 GiB 3 552 MB/s with 41 GB (the tail of every sub-batch's parse is idle time) */
    uint64_t sub_limit = 4096ull << 20;
    if (const char *e = getenv("ZSC_HIP_SUBBATCH_MB"))
        sub_limit = (uint64_t)atoll(e) << 20;
    if (sub_limit < (1ull << 20))
        sub_limit = 1ull << 20;
    if (runs)
        sub_limit = ~0ull; /* the caller reads the block records back: one set of scratch arrays */

    /* cut into sub-batches and number tiles / block slots / symbol slots inside each */
    uint64_t max_tiles = 0, max_slots = 0, max_syms = 0, max_rank_span = 0, max_count = 0;
    uint32_t i = 0;
    while (i < count) {
        SubBatch sb;
        sb.first = i;
        uint64_t bytes = 0;
        while (i < count && (sb.count == 0 || bytes + source_lens[i] <= sub_limit)) {
            const uint32_t n = source_lens[i];
            if (n >= ZSC_HIP_MAX_BUFFER) {
                /* bit positions inside one stream are 32-bit (ZdBlockPlan.bit_off, the layout and
                 * emit kernels): a stream must stay below 2^32 bits */
                ZSC_WARN2("zsc_hip: buffer %u has %u bytes; one buffer (or one run of sections) is limited "
                          "to 535 822 335 bytes.", i, n);
                delete pl;
                return Z_MEM_ERROR;
            }
            if (in_offsets[i] & 15u || out_offsets[i] & 15u) {
                ZSC_WARN1("zsc_hip: buffer %u is not 16-byte aligned in the batch.", i);
                delete pl;
                return Z_STREAM_ERROR;
            }
            ZdBuf &b = pl->bufs[i];
            memset(&b, 0, sizeof b);
            b.in_off = in_offsets[i];
            b.out_off = out_offsets[i];
            b.in_len = n;
            b.out_cap = out_caps[i];
            b.ntiles = n == 0 ? 1u : (uint32_t)(((uint64_t)n + ZD_TILE - 1) / ZD_TILE);
            b.tile0 = sb.ntiles;
            b.max_blocks = n / ((1u << (mem_level + 6)) - 1u) + 2;
            if (runs) {
                b.more = runs->more[i];
                b.n0 = runs->n0[i];
                b.sched_off = runs->sched_off[i];
                b.sched_n = runs->sched_n[i];
                b.seg_ok = runs->seg_ok[i];
                b.max_blocks += b.sched_n; /* a joint can cut a block */
            } else {
                b.n0 = n;
            }
            b.blk0 = sb.nslots;
            b.sym_off = sb.nsym_slots;
            b.rank_off = sb.nsym_slots; /* one u16 per (padded) input position */
            b.level = (uint32_t)level;
            b.wrap = (uint32_t)wrap;
            b.strategy = (uint32_t)strategy;
            b.wbits = (uint32_t)wbits;
            sb.ntiles += b.ntiles;
            sb.nslots += b.max_blocks;
            sb.nsym_slots += ((uint64_t)n + 64u) & ~63ull;
            bytes += n;
            sb.count++;
            i++;
        }
        max_tiles = std::max<uint64_t>(max_tiles, sb.ntiles);
        max_slots = std::max<uint64_t>(max_slots, sb.nslots);
        max_syms = std::max<uint64_t>(max_syms, sb.nsym_slots);
        max_rank_span = std::max<uint64_t>(max_rank_span, sb.nsym_slots + 64);
        max_count = std::max<uint64_t>(max_count, sb.count);
        pl->subs.push_back(std::move(sb));
    }

    pl->use_seg = getenv("ZSC_HIP_NO_SEG") == nullptr && kLevels[level].slow;
    /* per-sub-batch descriptor arrays */
    for (SubBatch &sb : pl->subs) {
        std::vector<uint32_t> tile_owner(sb.ntiles), blk_owner(sb.nslots), order(sb.count);
        for (uint32_t k = 0; k < sb.count; k++) {
            const ZdBuf &b = pl->bufs[sb.first + k];
            for (uint32_t t = 0; t < b.ntiles; t++)
                tile_owner[b.tile0 + t] = k;
            for (uint32_t s = 0; s < b.max_blocks; s++)
                blk_owner[b.blk0 + s] = k;
            order[k] = k;
        }
        /* buffers longer than this go to the segmented parser.  Measured on the x4096 batch: 18 432
         * (round 1: what fits the small rings stays with the wave-per-buffer parser) 2 843 + 37 ms of
         * parsing, 8 192: 2 853 + 13 ms, 3 072: 2 853 + 0 ms -- a workgroup per 4 KiB file still
         * beats a wave per file */
        static const uint32_t seg_min = getenv("ZSC_HIP_SEG_MIN") ? (uint32_t)atoi(getenv("ZSC_HIP_SEG_MIN")) : 3072u;
        pl->table_min = seg_min;
        /* longest first; those the segmented parser may take come first */
        auto seg_able = [&](uint32_t k) {
            const ZdBuf &b = pl->bufs[sb.first + k];
            return pl->use_seg && b.in_len > seg_min && (b.sched_n == 0 || b.seg_ok);
        };
        auto klass = [&](uint32_t k) { return seg_able(k) ? 1 : 2; };
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) {
            const int ka = klass(a), kc = klass(c);
            if (ka != kc)
                return ka < kc;
            return pl->bufs[sb.first + a].in_len > pl->bufs[sb.first + c].in_len;
        });
        {
            const bool full_only = getenv("ZSC_HIP_FULL_RING") != nullptr;
            uint32_t k = 0;
            auto len_at = [&](uint32_t idx) { return pl->bufs[sb.first + order[idx]].in_len; };
            while (k < sb.count && seg_able(order[k]))
                k++;
            sb.cseg = k;
            while (k < sb.count && (full_only || len_at(k) > 18432u))
                k++;
            sb.c36 = k;
            while (k < sb.count && len_at(k) > 10240u)
                k++;
            sb.c16 = k;
            while (k < sb.count && len_at(k) > 6144u)
                k++;
            sb.c8 = k;
        }
        bool ok = sb.d_bufs.ensure(sizeof(ZdBuf) * sb.count) &&
                  sb.d_tile_owner.ensure(4ull * std::max(1u, sb.ntiles)) &&
                  sb.d_blk_owner.ensure(4ull * std::max(1u, sb.nslots)) &&
                  sb.d_order.ensure(4ull * sb.count);
        if (!ok) {
            zsc_hip_deflate_plan_destroy(pl);
            return Z_MEM_ERROR;
        }
        HIP_TRY(hipMemcpy(sb.d_bufs.p, &pl->bufs[sb.first], sizeof(ZdBuf) * sb.count,
                          hipMemcpyHostToDevice),
                { zsc_hip_deflate_plan_destroy(pl); return Z_MEM_ERROR; });
        HIP_TRY(hipMemcpy(sb.d_tile_owner.p, tile_owner.data(), 4ull * sb.ntiles,
                          hipMemcpyHostToDevice),
                { zsc_hip_deflate_plan_destroy(pl); return Z_MEM_ERROR; });
        HIP_TRY(hipMemcpy(sb.d_blk_owner.p, blk_owner.data(), 4ull * sb.nslots,
                          hipMemcpyHostToDevice),
                { zsc_hip_deflate_plan_destroy(pl); return Z_MEM_ERROR; });
        HIP_TRY(hipMemcpy(sb.d_order.p, order.data(), 4ull * sb.count, hipMemcpyHostToDevice),
                { zsc_hip_deflate_plan_destroy(pl); return Z_MEM_ERROR; });
        pl->scratch_bytes += sb.d_bufs.bytes + sb.d_tile_owner.bytes + sb.d_blk_owner.bytes +
                             sb.d_order.bytes;
    }

    if (runs && runs->nsched) {
        if (!pl->d_sched.ensure(sizeof(ZdSched) * runs->nsched) ||
            hipMemcpy(pl->d_sched.p, runs->sched, sizeof(ZdSched) * runs->nsched,
                      hipMemcpyHostToDevice) != hipSuccess) {
            zsc_hip_deflate_plan_destroy(pl);
            return Z_MEM_ERROR;
        }
        pl->scratch_bytes += pl->d_sched.bytes;
    }
    uint64_t max_seg = 0;
    for (const SubBatch &sb : pl->subs)
        max_seg = std::max<uint64_t>(max_seg, sb.cseg);
    if (pl->use_seg && max_seg) {
        if (!pl->d_seg_tok.ensure(max_seg * SG_SCRATCH_WORDS * 4ull)) {
            zsc_hip_deflate_plan_destroy(pl);
            return Z_MEM_ERROR;
        }
        pl->scratch_bytes += pl->d_seg_tok.bytes;
    }

    /* shared scratch: sorted (4 B/position), tmp aliased with the symbol stream
     * (4 B/position, tmp is dead once the sort kernel ends), rank (2 B/position,
     * indexed by input offset), dir (2 B/position), block records + plans */
    const uint64_t tile_words = max_tiles * ZD_TILE;
    bool ok = pl->d_sorted.ensure(tile_words * 4) &&
              pl->d_tmp_syms.ensure(std::max<uint64_t>(tile_words, max_syms) * 4) &&
              pl->d_rank.ensure(max_rank_span * 2) && pl->d_hib.ensure(max_rank_span * 2) &&
              pl->d_cnt.ensure(max_rank_span * 4) &&
              pl->d_dir.ensure(max_tiles * ZD_DIR_STRIDE * 2) &&
              pl->d_recs.ensure(max_slots * sizeof(ZdBlockRec)) &&
              pl->d_plans.ensure(max_slots * sizeof(ZdBlockPlan)) &&
              pl->d_pout.ensure(max_count * sizeof(ZdParseOut)) &&
              pl->d_res.ensure((uint64_t)std::max(1u, count) * sizeof(ZdResult));
    /* the match table: levels 4-9 with the default window and hash size, for the buffers the
     * segmented parser takes; the match-finding strategies (default, filtered, fixed) share it */
    /* Off unless asked for (ZSC_HIP_TABLE=1): measured on the MI355X the table
     * kernel costs what the hops it buys save -- 23 ms per 403 MB of text for 40 % of the loop tops. */
    pl->use_table = getenv("ZSC_HIP_TABLE") != nullptr && pl->use_seg && max_seg != 0 && wbits == 15 &&
                    mem_level == 8 && strategy != Z_HUFFMAN_ONLY && strategy != Z_RLE;
    if (const char *e = getenv("ZSC_HIP_STAIR_MIN")) /* (debugging / tuning aid) */
        pl->stair_min = (uint32_t)atoi(e);
    if (const char *e = getenv("ZSC_HIP_TABLE_CAP")) /* (debugging / tuning aid) */
        pl->table_cap = (uint32_t)atoi(e);
    if (ok && pl->use_table)
        ok = pl->d_r2.ensure(max_rank_span * 4);
    if (!ok) {
        zsc_hip_deflate_plan_destroy(pl);
        return Z_MEM_ERROR;
    }
    pl->scratch_bytes += pl->d_r2.bytes;
    pl->scratch_bytes += pl->d_sorted.bytes + pl->d_tmp_syms.bytes + pl->d_rank.bytes + pl->d_hib.bytes + pl->d_cnt.bytes +
                         pl->d_dir.bytes + pl->d_recs.bytes + pl->d_plans.bytes +
                         pl->d_pout.bytes + pl->d_res.bytes;
    *plan_out = pl;
    return Z_OK;
}

extern "C" void zsc_hip_deflate_plan_profile(zsc_hip_deflate_plan *plan, I32 enable)
{
    ZSC_ASSERT(plan != Z_NULL);
    plan->profile = enable != 0;
    plan->events_used = 0; /* a new measurement window starts */
    plan->profiled_runs = 0;
}

extern "C" ZlibReturn zsc_hip_deflate_plan_run(zsc_hip_deflate_plan *pl, const void *d_input,
                                               void *d_output, void *hip_stream)
{
    DeviceScope scope;
    ZSC_ASSERT(pl != Z_NULL);
    ZSC_ASSERT(d_input != Z_NULL);
    ZSC_ASSERT(d_output != Z_NULL);
    hipStream_t st = (hipStream_t)hip_stream;
    pl->last_stream = st;
    (void)hipGetLastError(); /* an error some earlier call on this thread left behind is not this run's */
    const uint8_t *in = (const uint8_t *)d_input;
    uint8_t *out = (uint8_t *)d_output;
    ZdLevel cfg = kLevels[pl->level];
    cfg.wsize = 1u << pl->wbits;                 /* reference src/deflate.c:343-346 */
    cfg.max_dist = cfg.wsize - ZD_MIN_LOOKAHEAD; /* MAX_DIST, include/zsc/deflate.h:304 */
    cfg.sym_cap = (1u << (pl->mem_level + 6)) - 1u; /* lit_bufsize - 1, :362 + deflate.h:338-354 */
    cfg.hbits = (uint32_t)pl->mem_level + 7u;

    const size_t nev = pl->events_used + pl->subs.size() * ZSC_HIP_NKERNELS;
    if (pl->profile) {
        while (pl->events.size() < nev) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e), return Z_MEM_ERROR);
            pl->events.push_back(e);
        }
        pl->profiled_runs++;
    }
    auto mark = [&]() {
        if (pl->profile)
            (void)hipEventRecord(pl->events[pl->events_used++], st);
    };

    for (SubBatch &sb : pl->subs) {
        const ZdBuf *bufs = (const ZdBuf *)sb.d_bufs.p;
        ZdResult *res = (ZdResult *)pl->d_res.p + sb.first;
        uint32_t *sorted = (uint32_t *)pl->d_sorted.p;
        uint32_t *tmp_syms = (uint32_t *)pl->d_tmp_syms.p;
        uint16_t *rank = (uint16_t *)pl->d_rank.p;
        uint16_t *hib = (uint16_t *)pl->d_hib.p;
        uint32_t *cnt = (uint32_t *)pl->d_cnt.p;
        uint16_t *dir = (uint16_t *)pl->d_dir.p;
        ZdBlockRec *recs = (ZdBlockRec *)pl->d_recs.p;
        ZdBlockPlan *plans = (ZdBlockPlan *)pl->d_plans.p;
        ZdParseOut *pout = (ZdParseOut *)pl->d_pout.p;

        mark();
        hipLaunchKernelGGL(k_checksum, dim3(sb.count), dim3(64), 0, st, in, bufs, res, sb.count);
        mark();
        /* Z_HUFFMAN_ONLY / Z_RLE look at no earlier data than the previous byte: no chains */
        const bool simple = pl->strategy == (uint32_t)Z_HUFFMAN_ONLY || pl->strategy == (uint32_t)Z_RLE;
        const bool link_in_parser = !simple && cfg.slow && sb.cseg == sb.count && !pl->use_table &&
                                    getenv("ZSC_HIP_LINK_KERNEL") == nullptr; /* (the variable: A/B and debugging) */
        if (!simple) {
            hipLaunchKernelGGL(k_hash_sort, dim3(sb.ntiles), dim3(HS_WAVES * 64), 0, st, in, bufs,
                               (const uint32_t *)sb.d_tile_owner.p, sorted, tmp_syms, rank, dir,
                               sb.ntiles);
            /* chain lengths and the links into the previous tile, for every position -- unless every buffer of
             * the batch goes to the segmented parser, which works them out for the positions it visits */
            if (!link_in_parser)
                hipLaunchKernelGGL(k_link_prev, dim3(sb.ntiles), dim3(HS_WAVES * 64), 0, st, in, bufs,
                                   (const uint32_t *)sb.d_tile_owner.p, (const uint16_t *)dir,
                                   (const uint16_t *)rank, hib, cnt, sb.ntiles);
        }
        mark();
        if (!simple) {
            if (pl->use_table && sb.cseg > 0)
                hipLaunchKernelGGL(k_match_table, dim3(sb.ntiles), dim3(MT_WAVES * 64), 0, st, in, bufs,
                                   (const uint32_t *)sb.d_tile_owner.p, (const uint32_t *)sorted,
                                   (const uint16_t *)rank, (const uint16_t *)hib, (const uint32_t *)cnt,
                                   (uint32_t *)pl->d_r2.p, cfg, pl->table_min,
                                   pl->table_cap, sb.ntiles);
        }
        mark();
        if (simple) {
            hipLaunchKernelGGL(k_parse_simple, dim3(sb.count), dim3(64), 0, st, in, bufs, tmp_syms,
                               recs, pout, (const ZdSched *)pl->d_sched.p, cfg, sb.count);
            mark();
        } else if (cfg.slow) {
#define ZSC_LAUNCH_PARSE(LT, FIRST, COUNT)                                                       \
    if ((COUNT) > 0)                                                                             \
    hipLaunchKernelGGL(k_parse<LT>, dim3(COUNT), dim3(64), 0, st, in, bufs,                      \
                       (const uint32_t *)sb.d_order.p, (const uint32_t *)sorted,                 \
                       (const uint16_t *)rank, (const uint16_t *)hib, tmp_syms, recs, pout,      \
                       (const ZdSched *)pl->d_sched.p, cfg, (uint32_t)(FIRST), (uint32_t)(COUNT))
            if (sb.cseg > 0) {
                auto kern = (pl->wbits == 15 && pl->mem_level == 8)
                                ? (pl->use_table ? k_parse_seg<false, true, 0>
                                                 : pl->level == 6 && pl->strategy != (uint32_t)Z_FILTERED ? k_parse_seg<false, false, 6>
                                                                                                          : k_parse_seg<false, false, 0>)
                                : k_parse_seg<true, false, 0>;
                hipLaunchKernelGGL(kern, dim3(sb.cseg), dim3(SG_W * 64), 0, st, in, bufs,
                                   (const uint32_t *)sb.d_order.p, (const uint32_t *)sorted,
                                   (const uint16_t *)rank, (const uint16_t *)hib,
                                   (const uint32_t *)cnt, link_in_parser ? (const uint16_t *)dir : nullptr,
                                   pl->use_table ? (const uint32_t *)pl->d_r2.p : nullptr, tmp_syms, recs,
                                   pout, (uint32_t *)pl->d_seg_tok.p,
                                   (const ZdSched *)pl->d_sched.p, cfg, pl->stair_min, 0u, sb.cseg);
            }
            if (pl->d_sched.p) { /* runs with joints: the parsers keep a hole map (lz_parse.h) */
                ZSC_LAUNCH_PARSE(LzLdsJ, sb.cseg, sb.c36 - sb.cseg);
                mark();
                ZSC_LAUNCH_PARSE(LzLdsJ16k, sb.c36, sb.c16 - sb.c36);
                ZSC_LAUNCH_PARSE(LzLdsJ8k, sb.c16, sb.c8 - sb.c16);
                ZSC_LAUNCH_PARSE(LzLdsJ4k, sb.c8, sb.count - sb.c8);
            } else {
                ZSC_LAUNCH_PARSE(LzLds, sb.cseg, sb.c36 - sb.cseg);
                mark();
                ZSC_LAUNCH_PARSE(LzLds16k, sb.c36, sb.c16 - sb.c36);
                ZSC_LAUNCH_PARSE(LzLds8k, sb.c16, sb.c8 - sb.c16);
                ZSC_LAUNCH_PARSE(LzLds4k, sb.c8, sb.count - sb.c8);
            }
#undef ZSC_LAUNCH_PARSE
        } else
            hipLaunchKernelGGL(getenv("ZSC_HIP_FAST_RING") ? k_parse_fast<LzLdsFast> : k_parse_fast<LzLdsFastG>,
                               dim3(sb.count), dim3(64), 0, st, in, bufs,
                               (const uint32_t *)sb.d_order.p, (const uint32_t *)sorted,
                               (const uint16_t *)rank, (const uint16_t *)hib, tmp_syms, recs,
                               pout, (const ZdSched *)pl->d_sched.p, cfg, sb.count);
        if (!simple && !cfg.slow)
            mark(); /* levels 1-3: one parse kernel for every length, the short-buffer slot stays empty */
        mark();
        hipLaunchKernelGGL(k_huff_plan, dim3(sb.nslots), dim3(64), 0, st, bufs,
                           (const uint32_t *)sb.d_blk_owner.p, (const uint32_t *)tmp_syms,
                           (const ZdBlockRec *)recs, (const ZdParseOut *)pout, plans, sb.nslots);
        mark();
        hipLaunchKernelGGL(k_layout, dim3((sb.count + 63) / 64), dim3(64), 0, st, bufs,
                           (const ZdParseOut *)pout, (const ZdBlockRec *)recs, plans, res, out,
                           sb.count);
        mark();
        hipLaunchKernelGGL(k_emit, dim3(sb.nslots), dim3(64), 0, st, in, bufs,
                           (const uint32_t *)sb.d_blk_owner.p, (const uint32_t *)tmp_syms,
                           (const ZdBlockRec *)recs, (const ZdParseOut *)pout,
                           (const ZdBlockPlan *)plans, out, sb.nslots);
        mark();
    }
    HIP_TRY(hipGetLastError(), return Z_STREAM_ERROR);
    return Z_OK;
}

extern "C" ZlibReturn zsc_hip_deflate_plan_results(zsc_hip_deflate_plan *pl, U32 *dest_lens,
                                                   I32 *statuses)
{
    DeviceScope scope;
    ZSC_ASSERT(pl != Z_NULL);
    HIP_TRY(hipStreamSynchronize(pl->last_stream), return Z_STREAM_ERROR);
    std::vector<ZdResult> res(pl->count);
    if (pl->count)
        HIP_TRY(hipMemcpy(res.data(), pl->d_res.p, sizeof(ZdResult) * pl->count,
                          hipMemcpyDeviceToHost),
                return Z_STREAM_ERROR);
    for (uint32_t i = 0; i < pl->count; i++) {
        if (dest_lens)
            dest_lens[i] = res[i].out_len;
        if (statuses)
            statuses[i] = res[i].status;
    }
    if (pl->profile && pl->profiled_runs) {
        /* mean per run over every run since profiling was switched on */
        for (int k = 0; k < ZSC_HIP_NKERNELS; k++)
            pl->times[k] = 0.f;
        const int NE = ZSC_HIP_NKERNELS; /* marks per sub-batch: NE-1 intervals + the whole pass */
        const size_t per_run = pl->subs.size() * NE;
        for (uint32_t r = 0; r < pl->profiled_runs; r++) {
            const size_t e0 = (size_t)r * per_run;
            for (size_t s = 0; s < pl->subs.size(); s++) {
                for (int k = 0; k < NE - 1; k++) {
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, pl->events[e0 + s * NE + k],
                                              pl->events[e0 + s * NE + k + 1]);
                    pl->times[k] += ms;
                }
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, pl->events[e0], pl->events[e0 + per_run - 1]);
            pl->times[NE - 1] += ms;
        }
        for (int k = 0; k < ZSC_HIP_NKERNELS; k++)
            pl->times[k] /= (float)pl->profiled_runs;
    }
    return Z_OK;
}

extern "C" ZlibReturn zsc_hip_deflate_plan_times(zsc_hip_deflate_plan *pl, float *ms_out)
{
    DeviceScope scope;
    ZSC_ASSERT(pl != Z_NULL);
    ZSC_ASSERT(ms_out != Z_NULL);
    for (int k = 0; k < ZSC_HIP_NKERNELS; k++)
        ms_out[k] = pl->times[k];
    return pl->profile ? Z_OK : Z_STREAM_ERROR;
}

extern "C" uint64_t zsc_hip_deflate_plan_scratch_bytes(const zsc_hip_deflate_plan *pl)
{
    return pl ? pl->scratch_bytes : 0;
}

extern "C" U32 zsc_hip_deflate_plan_sub_batches(const zsc_hip_deflate_plan *pl)
{
    return pl ? (U32)pl->subs.size() : 0;
}

extern "C" void zsc_hip_deflate_plan_destroy(zsc_hip_deflate_plan *pl)
{
    DeviceScope scope;
    if (!pl)
        return;
    (void)hipStreamSynchronize(pl->last_stream); /* the blocks go back to the cache, not to hipFree */
    for (SubBatch &sb : pl->subs) {
        sb.d_bufs.release();
        sb.d_tile_owner.release();
        sb.d_blk_owner.release();
        sb.d_order.release();
    }
    pl->d_sorted.release();
    pl->d_tmp_syms.release();
    pl->d_rank.release();
    pl->d_hib.release();
    pl->d_cnt.release();
    pl->d_seg_tok.release();
    pl->d_r2.release();
    pl->d_sched.release();
    pl->d_dir.release();
    pl->d_recs.release();
    pl->d_plans.release();
    pl->d_pout.release();
    pl->d_res.release();
    for (hipEvent_t e : pl->events)
        (void)hipEventDestroy(e);
    delete pl;
}

/* ---- level 0 --------------------------------------------------------------------- */

namespace {

/* Where the pieces of a level-0 stream go.  deflate_stored (reference src/deflate.c:1679-1880)
 * sizes its stored blocks by the output space of the moment, and zsc_compress hands that
 * space out in slices of max_block_len next to input sections of max_block_len
 * (src/zsc_compress.c:121-138), so the layout is found by following those calls -- pure
 * arithmetic on lengths, the bytes themselves never matter.  One StoreSim per buffer. */
struct StoreSim {
    uint32_t w_size = 0, pending_buf_size = 0, max_block_len = 0, source_len = 0;
    int wrap = 1;
    std::vector<ZdStorePiece> *pieces = nullptr;
    uint64_t in_base = 0, out_base = 0;
    uint32_t buf = 0, hdr_arg = 0;
    uint32_t hdr_len = 0; /* a caller-supplied gzip member header of this length (0: the plain one) */
    /* the stream */
    uint32_t produced = 0, delivered = 0, avail_out = 0;
    uint32_t given = 0, st_read = 0, st_emit = 0, st_strstart = 0, st_block_start = 0;
    bool header_done = false, finishing = false, trailer_done = false;

    void piece(uint32_t kind, uint32_t len, uint32_t arg, uint32_t bytes)
    {
        ZdStorePiece pc;
        pc.src_off = in_base + st_emit;
        pc.dst_off = out_base + produced;
        pc.len = len;
        pc.kind = kind;
        pc.arg = arg;
        pc.buf = buf;
        pieces->push_back(pc);
        produced += bytes;
    }
    void stored_block(uint32_t len, int last)
    {
        piece(0, len, (uint32_t)last, 5u + len);
        st_emit += len;
    }
    void flush_pending()
    {
        const uint32_t have = produced - delivered, len = have < avail_out ? have : avail_out;
        delivered += len;
        avail_out -= len;
    }
    /* 0 need_more, 1 block_done, 2 finish_started, 3 finish_done */
    int deflate_stored(bool finish)
    {
        const uint32_t window_size = 2u * w_size;
        uint32_t min_block = std::min(pending_buf_size - 5u, w_size);
        uint32_t avail_in = given - st_read;
        const uint32_t used0 = avail_in;
        uint32_t len, left, have;
        int last = 0;
        do {
            len = 65535u;
            have = 5u;
            if (avail_out < have)
                break;
            have = avail_out - have;
            left = st_strstart - st_block_start;
            len = std::min(len, left + avail_in);
            len = std::min(len, have);
            if (len < min_block && ((len == 0 && !finish) || len != left + avail_in))
                break;
            last = finish && len == left + avail_in;
            const uint32_t from_window = std::min(left, len);
            stored_block(len, last);
            st_block_start += from_window;
            st_read += len - from_window;
            avail_in -= len - from_window;
            delivered += 5u + len; /* header through pending, bytes straight to next_out */
            avail_out -= 5u + len;
        } while (!last);
        const uint32_t used = used0 - avail_in;
        if (used) {
            if (used >= w_size) {
                st_strstart = w_size;
            } else {
                if (window_size - st_strstart <= used)
                    st_strstart -= w_size;
                st_strstart += used;
            }
            st_block_start = st_strstart;
        }
        if (last)
            return 3;
        if (!finish && avail_in == 0 && st_strstart == st_block_start)
            return 1;
        have = window_size - st_strstart - 1u;
        if (avail_in > have && st_block_start >= w_size) {
            st_block_start -= w_size;
            st_strstart -= w_size;
            have += w_size;
        }
        have = std::min(have, avail_in);
        if (have) {
            st_read += have;
            avail_in -= have;
            st_strstart += have;
        }
        have = std::min(pending_buf_size - 5u, 65535u);
        min_block = std::min(have, w_size);
        left = st_strstart - st_block_start;
        if (left >= min_block || ((left || finish) && avail_in == 0 && left <= have)) {
            len = std::min(left, have);
            last = finish && avail_in == 0 && len == left;
            stored_block(len, last);
            st_block_start += len;
            flush_pending();
        }
        return last ? 2 : 0;
    }
    /* one deflate() call: Z_OK, 1 = Z_STREAM_END, Z_BUF_ERROR (src/deflate.c:964-1295) */
    int deflate(bool finish)
    {
        if (avail_out == 0)
            return Z_BUF_ERROR;
        if (produced != delivered) {
            flush_pending();
            if (avail_out == 0)
                return Z_OK;
        }
        if (!header_done) {
            header_done = true;
            if (wrap) {
                if (hdr_len)
                    piece(4, 0, 0, hdr_len); /* room only: zsc_api.c writes the caller's header */
                else
                    piece(2, 0, hdr_arg, wrap == 1 ? 2u : 10u);
                flush_pending();
                if (produced != delivered)
                    return Z_OK;
            }
        }
        if (!finishing) {
            const int bs = deflate_stored(finish);
            if (bs == 2 || bs == 3)
                finishing = true;
            if (bs == 0 || bs == 2)
                return Z_OK;
            if (bs == 1) {
                piece(1, 0, 0, 5u);
                st_strstart = st_block_start = 0;
                flush_pending();
                return Z_OK;
            }
        }
        if (!finish)
            return Z_OK;
        if (wrap == 0)
            return 1;
        if (!trailer_done) {
            trailer_done = true;
            piece(3, 0, 0, wrap == 1 ? 4u : 8u);
            flush_pending();
            return produced != delivered ? Z_OK : 1;
        }
        return 1;
    }
    /* the wrapper's loop, src/zsc_compress.c:121-138; returns the call's ZlibReturn */
    int run(uint32_t dest_cap)
    {
        uint32_t left_dest = dest_cap, left_src = source_len;
        int err = Z_OK;
        while (err == Z_OK) {
            if (avail_out == 0) {
                avail_out = std::min(left_dest, max_block_len);
                left_dest -= avail_out;
            }
            if (st_read == given) {
                const uint32_t take = std::min(left_src, max_block_len);
                given += take;
                left_src -= take;
            }
            err = deflate(left_src == 0);
        }
        return err == 1 ? Z_OK : err;
    }
};

} // namespace

extern "C" ZlibReturn zsc_hip_store_batch(U32 count, const U8 *const *sources, const U32 *source_lens,
                                          const U32 *max_block_lens, U8 *const *dests,
                                          U32 *dest_lens, I32 *statuses, I32 window_bits,
                                          I32 mem_level, U32 gzip_header_len)
{
    DeviceScope scope;
    ZSC_ASSERT(sources != Z_NULL);
    ZSC_ASSERT(source_lens != Z_NULL);
    ZSC_ASSERT(max_block_lens != Z_NULL);
    ZSC_ASSERT(dests != Z_NULL);
    ZSC_ASSERT(dest_lens != Z_NULL);
    if (count == 0)
        return Z_OK;
    if (zsc_hip_init(-1) != Z_OK)
        return Z_STREAM_ERROR;
    int wrap = 1, wbits = 15;
    if (!offloadable(1, window_bits, mem_level, Z_DEFAULT_STRATEGY, &wrap, &wbits))
        return Z_STREAM_ERROR;
    /* zlib header of level 0: level_flags 0 (src/deflate.c:1031-1049); gzip XFL 4 (:1075-1078) */
    uint32_t zh = (8u + (((uint32_t)wbits - 8u) << 4)) << 8;
    zh += 31u - zh % 31u;
    std::vector<ZdStorePiece> pieces;
    std::vector<ZdBuf> bufs(count);
    std::vector<uint32_t> give(count), stat(count);
    uint64_t in_bytes = 0, out_bytes = 0;
    for (U32 i = 0; i < count; i++) {
        ZSC_ASSERT(max_block_lens[i] != 0);
        StoreSim sim;
        sim.w_size = 1u << wbits;
        sim.pending_buf_size = (1u << (mem_level + 6)) * 4u; /* lit_bufsize * 4, src/deflate.c:362 */
        sim.max_block_len = max_block_lens[i];
        sim.source_len = source_lens[i];
        sim.wrap = wrap;
        sim.pieces = &pieces;
        sim.in_base = in_bytes;
        sim.out_base = out_bytes;
        sim.buf = i;
        sim.hdr_arg = wrap == 1 ? zh : 4u;
        sim.hdr_len = wrap == 2 ? gzip_header_len : 0u;
        stat[i] = (uint32_t)sim.run(dest_lens[i]);
        give[i] = sim.delivered;
        memset(&bufs[i], 0, sizeof(ZdBuf));
        bufs[i].in_off = in_bytes;
        bufs[i].out_off = out_bytes;
        bufs[i].in_len = source_lens[i];
        bufs[i].wrap = (uint32_t)wrap;
        in_bytes += ((uint64_t)source_lens[i] + 15u) & ~15ull;
        out_bytes += ((uint64_t)sim.produced + 15u) & ~15ull;
    }
    DevBuf d_in, d_out, d_pieces, d_bufs, d_res;
    ZlibReturn rc = Z_OK;
    if (!d_in.ensure(in_bytes + 64) || !d_out.ensure(out_bytes + 64) ||
        !d_pieces.ensure(sizeof(ZdStorePiece) * std::max<size_t>(1, pieces.size())) ||
        !d_bufs.ensure(sizeof(ZdBuf) * count) || !d_res.ensure(sizeof(ZdResult) * count))
        rc = Z_MEM_ERROR;
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        ZSC_ASSERT(sources[i] != Z_NULL);
        if (source_lens[i] && hipMemcpy((uint8_t *)d_in.p + bufs[i].in_off, sources[i], source_lens[i],
                                        hipMemcpyHostToDevice) != hipSuccess)
            rc = Z_STREAM_ERROR;
    }
    if (rc == Z_OK &&
        (hipMemcpy(d_bufs.p, bufs.data(), sizeof(ZdBuf) * count, hipMemcpyHostToDevice) != hipSuccess ||
         (!pieces.empty() && hipMemcpy(d_pieces.p, pieces.data(), sizeof(ZdStorePiece) * pieces.size(),
                                       hipMemcpyHostToDevice) != hipSuccess)))
        rc = Z_STREAM_ERROR;
    if (rc == Z_OK) {
        hipLaunchKernelGGL(k_checksum, dim3(count), dim3(64), 0, nullptr, (const uint8_t *)d_in.p,
                           (const ZdBuf *)d_bufs.p, (ZdResult *)d_res.p, count);
        if (!pieces.empty())
            hipLaunchKernelGGL(k_store, dim3((uint32_t)pieces.size()), dim3(256), 0, nullptr,
                               (const uint8_t *)d_in.p, (uint8_t *)d_out.p,
                               (const ZdStorePiece *)d_pieces.p, (const ZdBuf *)d_bufs.p,
                               (const ZdResult *)d_res.p, (uint32_t)pieces.size());
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess)
            rc = Z_STREAM_ERROR;
    }
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        ZSC_ASSERT(dests[i] != Z_NULL);
        if (give[i] && hipMemcpy(dests[i], (uint8_t *)d_out.p + bufs[i].out_off, give[i],
                                 hipMemcpyDeviceToHost) != hipSuccess)
            rc = Z_STREAM_ERROR;
        dest_lens[i] = give[i];
        if (statuses)
            statuses[i] = (I32)stat[i];
    }
    d_in.release();
    d_out.release();
    d_pieces.release();
    d_bufs.release();
    d_res.release();
    return rc;
}

/* ---- levels 1-9 with source_len > max_block_len (sections.h) ----------------------- */

namespace {

/* runs the kernels for the runs sections.h wants parsed (one plan per round) and keeps every
 * round's compressed bytes until the streams are put together */
struct HipSecRunner {
    const uint8_t *d_src = nullptr; /* the streams' input, stream i at src_off[i] */
    const uint64_t *src_off = nullptr;
    I32 level = 6, mem_level = 8;
    int wbits = 15;
    ZlibStrategy strategy = Z_DEFAULT_STRATEGY;
    struct Round {
        DevBuf d_in, d_out;
        std::vector<uint64_t> out_off;
    };
    std::vector<Round *> rounds;
    uint32_t parses = 0;
    ZlibReturn error = Z_OK;
    const std::vector<SecStream> *streams = nullptr; /* whose runs say which rounds' bytes are still in use */
    uint64_t parsed_bytes = 0, budget_bytes = ~0ull;  /* work budget: input bytes parsed over all rounds */
    uint64_t held_bytes = 0, held_peak = 0;           /* compressed bytes of the rounds kept */

    /* Bytes of a round are in use while a run that was parsed in it is still part of its stream (not
     * about to be parsed again: `jobs`, and not swallowed by the run before it) or a finished stream
     * is made of them.  Everything else is given back before the next round allocates: without this
     * a stream that needs many rounds (short sections, tiny blocks, incompressible data: every
     * section end a joint) held rounds x run bytes, hundreds of GB for a few MiB of input. */
    void release_dead_rounds(const std::vector<SecRun *> &jobs)
    {
        if (!streams)
            return;
        std::vector<uint8_t> live(rounds.size(), 0);
        for (const SecStream &st : *streams) {
            if (st.done) {
                for (const SecPiece &pc : st.pieces)
                    if ((pc.kind == SEC_PIECE_RUN || pc.kind == SEC_PIECE_TAIL) && pc.round < live.size())
                        live[pc.round] = 1;
                continue;
            }
            for (const auto &kv : st.runs) {
                const SecRun &r = kv.second;
                if (r.blocks.empty() || std::find(jobs.begin(), jobs.end(), &r) != jobs.end())
                    continue; /* never parsed yet, or about to be parsed again */
                if (r.round < live.size())
                    live[r.round] = 1;
            }
        }
        for (size_t k = 0; k < rounds.size(); k++) {
            if (rounds[k] && !live[k] && rounds[k]->d_out.p) {
                held_bytes -= rounds[k]->d_out.bytes;
                rounds[k]->d_out.release();
            }
        }
    }

    ~HipSecRunner()
    {
        for (Round *r : rounds) {
            r->d_in.release();
            r->d_out.release();
            delete r;
        }
    }

    int operator()(std::vector<SecRun *> &jobs, uint32_t round)
    {
        const auto t_begin = std::chrono::steady_clock::now();
        const U32 count = (U32)jobs.size();
        release_dead_rounds(jobs);
        for (const SecRun *r : jobs)
            parsed_bytes += r->n;
        if (parsed_bytes > budget_bytes) {
            ZSC_WARN3("zsc_hip: compressing these streams of sections needs more than %llu MB of parsing for "
                      "%u runs in round %u (sections so short, or blocks so small, that nearly every section end "
                      "lets the next section in): refused.",
                      (unsigned long long)(budget_bytes >> 20), count, round);
            return error = Z_STREAM_ERROR;
        }
        std::vector<U32> lens(count), caps(count), more(count), n0(count), soff(count), scnt(count), segok(count);
        std::vector<uint64_t> in_off(count), out_off(count);
        std::vector<ZdSched> sched;
        for (U32 j = 0; j < count; j++) {
            const SecRun &r = *jobs[j];
            lens[j] = r.n;
            more[j] = r.more ? 1u : 0u;
            n0[j] = r.n0;
            soff[j] = (U32)sched.size();
            scnt[j] = (U32)r.sched.size();
            segok[j] = sec_seg_ok(r) ? 1u : 0u;
            sched.insert(sched.end(), r.sched.begin(), r.sched.end());
        }
        uint64_t in_bytes = 0, out_bytes = 0;
        ZlibReturn rc = zsc_hip_deflate_plan_layout(count, lens.data(), level, -wbits, mem_level,
                                                    in_off.data(), out_off.data(), caps.data(),
                                                    &in_bytes, &out_bytes);
        if (rc != Z_OK)
            return error = rc;
        /* every joint can cut a block, a cut costs a few bytes; room for the marker's bits */
        out_bytes = 0;
        for (U32 j = 0; j < count; j++) {
            caps[j] += 16u * (scnt[j] + 1u) + 64u;
            out_off[j] = out_bytes;
            out_bytes += ((uint64_t)caps[j] + 16u + 15u) & ~15ull;
        }
        out_bytes += 64;
        Round *rd = new Round();
        rounds.resize(round + 1, nullptr);
        rounds[round] = rd;
        rd->out_off = out_off;
        if (!rd->d_out.ensure(out_bytes))
            return error = Z_MEM_ERROR;
        held_bytes += rd->d_out.bytes;
        held_peak = std::max(held_peak, held_bytes);

        /* the scratch arrays of a plan are ~16 bytes per input byte: groups of jobs, one plan each,
         * all writing into the round's output */
        uint64_t group_limit = 2048ull << 20;
        if (const char *e = getenv("ZSC_HIP_SECTIONS_GROUP_MB"))
            group_limit = std::max<uint64_t>(1, (uint64_t)atoll(e)) << 20;
        for (U32 g0 = 0; g0 < count && rc == Z_OK;) {
            U32 g1 = g0;
            uint64_t bytes = 0;
            while (g1 < count && (g1 == g0 || bytes + lens[g1] <= group_limit))
                bytes += lens[g1++];
            rc = run_group(jobs, round, g0, g1, lens, caps, more, n0, soff, scnt, segok, sched, out_off, rd);
            g0 = g1;
        }
        parses += count;
        if (getenv("ZSC_HIP_SECTIONS_LOG")) {
            uint64_t bytes = 0, joints = 0;
            for (U32 j = 0; j < count; j++)
                bytes += lens[j], joints += scnt[j];
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            fprintf(stderr, "zsc_hip sections: round %u: %u runs, %llu bytes, %llu joints, %.2f ms\n", round, count,
                    (unsigned long long)bytes, (unsigned long long)joints, ms);
        }
        return error = rc;
    }

    /* jobs [g0, g1) of a round: gather their input, run the kernels, read the block records back */
    ZlibReturn run_group(std::vector<SecRun *> &jobs, uint32_t round, U32 g0, U32 g1,
                         const std::vector<U32> &lens, const std::vector<U32> &caps,
                         const std::vector<U32> &more, const std::vector<U32> &n0,
                         const std::vector<U32> &soff, const std::vector<U32> &scnt,
                         const std::vector<U32> &segok, const std::vector<ZdSched> &sched,
                         const std::vector<uint64_t> &out_off, Round *rd)
    {
        const U32 count = g1 - g0;
        std::vector<uint64_t> in_off(count);
        uint64_t in_bytes = 0;
        for (U32 j = 0; j < count; j++) {
            in_off[j] = in_bytes;
            in_bytes += ((uint64_t)lens[g0 + j] + 15u) & ~15ull;
        }
        in_bytes += 64;
        PlanRuns pr = {more.data() + g0, n0.data() + g0, soff.data() + g0, scnt.data() + g0, segok.data() + g0,
                       sched.data(), (uint32_t)sched.size()};
        zsc_hip_deflate_plan *pl = nullptr;
        ZlibReturn rc = plan_create(&pl, count, lens.data() + g0, in_off.data(), out_off.data() + g0,
                                    caps.data() + g0, level, -wbits, mem_level, strategy, &pr);
        if (rc != Z_OK)
            return rc;
        DevBuf d_pieces;
        std::vector<ZdStorePiece> gather(count);
        for (U32 j = 0; j < count; j++) {
            ZdStorePiece &pc = gather[j];
            memset(&pc, 0, sizeof pc);
            pc.src_off = src_off[jobs[g0 + j]->stream] + jobs[g0 + j]->start;
            pc.dst_off = in_off[j];
            pc.len = lens[g0 + j];
            pc.kind = 6u;
        }
        if (!rd->d_in.ensure(in_bytes) || !d_pieces.ensure(sizeof(ZdStorePiece) * count))
            rc = Z_MEM_ERROR;
        if (rc == Z_OK && hipMemcpy(d_pieces.p, gather.data(), sizeof(ZdStorePiece) * count,
                                    hipMemcpyHostToDevice) != hipSuccess)
            rc = Z_STREAM_ERROR;
        if (rc == Z_OK) {
            hipLaunchKernelGGL(k_store, dim3(count), dim3(256), 0, nullptr, d_src, (uint8_t *)rd->d_in.p,
                               (const ZdStorePiece *)d_pieces.p, (const ZdBuf *)nullptr,
                               (const ZdResult *)nullptr, count);
            rc = zsc_hip_deflate_plan_run(pl, rd->d_in.p, rd->d_out.p, nullptr);
        }
        std::vector<U32> out_lens(count);
        std::vector<I32> stat(count);
        if (rc == Z_OK)
            rc = zsc_hip_deflate_plan_results(pl, out_lens.data(), stat.data());
        /* what the simulation needs of every block: where it ends, in input and in bits */
        const SubBatch &sb = pl->subs[0];
        std::vector<ZdBlockRec> recs(sb.nslots);
        std::vector<uint32_t> bit_off(sb.nslots);
        std::vector<ZdParseOut> pout(count);
        std::vector<ZdResult> res(count);
        DevBuf d_bits;
        if (rc == Z_OK && !d_bits.ensure(4ull * std::max(1u, sb.nslots)))
            rc = Z_MEM_ERROR;
        if (rc == Z_OK && sb.nslots)
            hipLaunchKernelGGL(k_bit_offs, dim3((sb.nslots + 255) / 256), dim3(256), 0, nullptr,
                               (const ZdBlockPlan *)pl->d_plans.p, (uint32_t *)d_bits.p, sb.nslots);
        if (rc == Z_OK &&
            (hipMemcpy(recs.data(), pl->d_recs.p, sizeof(ZdBlockRec) * sb.nslots, hipMemcpyDeviceToHost) != hipSuccess ||
             hipMemcpy(bit_off.data(), d_bits.p, 4ull * sb.nslots, hipMemcpyDeviceToHost) != hipSuccess ||
             hipMemcpy(pout.data(), pl->d_pout.p, sizeof(ZdParseOut) * count, hipMemcpyDeviceToHost) != hipSuccess ||
             hipMemcpy(res.data(), pl->d_res.p, sizeof(ZdResult) * count, hipMemcpyDeviceToHost) != hipSuccess))
            rc = Z_STREAM_ERROR;
        for (U32 j = 0; j < count && rc == Z_OK; j++) {
            SecRun &r = *jobs[g0 + j];
            const ZdBuf &b = pl->bufs[j];
            if (stat[j] != Z_OK || pout[j].nblocks > b.max_blocks) {
                ZSC_WARN2("zsc_hip: a run of sections came back with status %d (%u blocks).",
                          (int)stat[j], pout[j].nblocks);
                rc = Z_STREAM_ERROR;
                break;
            }
            r.blocks.clear();
            for (uint32_t k = 0; k < pout[j].nblocks; k++) {
                const ZdBlockRec &rec = recs[b.blk0 + k];
                SecBlock blk;
                blk.upto = rec.in_begin + rec.in_len;
                blk.end_bit = k + 1 < pout[j].nblocks ? bit_off[b.blk0 + k + 1] : res[j].bits;
                blk.wend = rec.wend;
                blk.at = rec.at;
                blk.cut = rec.cut;
                blk.last = rec.last;
                r.blocks.push_back(blk);
            }
            r.round = round;
            r.job = g0 + j;
        }
        d_pieces.release();
        d_bits.release();
        rd->d_in.release(); /* only the compressed bytes are needed later */
        zsc_hip_deflate_plan_destroy(pl);
        return rc;
    }
};

} // namespace

/* the streams are in device memory, stream i at d_src + src_off[i] (16-byte aligned); the finished
 * streams go to d_dst + dst_off[i], of which dest_caps[i] bytes may be used */
static ZlibReturn sections_on_device(U32 count, const uint8_t *d_src, const uint64_t *src_off,
                                     const U32 *source_lens, const U32 *max_block_lens,
                                     uint8_t *d_dst, const uint64_t *dst_off, const U32 *dest_caps,
                                     U32 *dest_lens, I32 *statuses, I32 level, I32 window_bits,
                                     I32 mem_level, ZlibStrategy strategy, U32 gzip_header_len)
{
    DeviceScope scope;
    if (zsc_hip_init(-1) != Z_OK)
        return Z_STREAM_ERROR;
    int wrap = 1, wbits = 15;
    if (!offloadable(level, window_bits, mem_level, strategy, &wrap, &wbits)) {
        ZSC_WARN4("zsc_hip: level %d / window_bits %d / mem_level %d / strategy %d is not "
                  "offloaded to the GPU yet (DESIGN.md, out of scope).",
                  level, window_bits, mem_level, (int)strategy);
        return Z_STREAM_ERROR;
    }
    if (level == Z_DEFAULT_COMPRESSION)
        level = 6;

    /* wrapper header as deflate() writes it (src/deflate.c:1031-1049, :1068-1082) */
    uint32_t zh = (8u + (((uint32_t)wbits - 8u) << 4)) << 8;
    const uint32_t lf = (strategy >= Z_HUFFMAN_ONLY || level < 2) ? 0u : level < 6 ? 1u : level == 6 ? 2u : 3u;
    zh |= lf << 6;
    zh += 31u - zh % 31u;
    const uint32_t xfl = level == 9 ? 2u : (strategy >= Z_HUFFMAN_ONLY || level < 2) ? 4u : 0u;

    /* the host keeps a record per section: refuse what would not fit there, loudly */
    uint64_t nsections = 0;
    for (U32 i = 0; i < count; i++) {
        ZSC_ASSERT(max_block_lens[i] != 0);
        nsections += source_lens[i] / max_block_lens[i] + 1u;
    }
    if (nsections > (64ull << 20)) {
        ZSC_WARN1("zsc_hip: %llu sections in one call are more than the host side keeps track of.",
                  (unsigned long long)nsections);
        return Z_MEM_ERROR;
    }
    std::vector<SecStream> streams(count);
    std::vector<ZdBuf> sbufs(count);
    for (U32 i = 0; i < count; i++) {
        ZSC_ASSERT(max_block_lens[i] != 0);
        if (src_off[i] & 15u) {
            ZSC_WARN1("zsc_hip: stream %u is not 16-byte aligned in the batch.", i);
            return Z_STREAM_ERROR;
        }
        SecStream &s = streams[i];
        s.source_len = source_lens[i];
        s.max_block_len = max_block_lens[i];
        s.dest_cap = dest_caps[i];
        s.wrap = wrap;
        s.hdr_len = wrap == 1 ? 2u : wrap == 2 ? (gzip_header_len ? gzip_header_len : 10u) : 0u;
        s.need = strategy == Z_HUFFMAN_ONLY ? 1u : strategy == Z_RLE ? ZD_MAX_MATCH + 1u : ZD_MIN_LOOKAHEAD;
        memset(&sbufs[i], 0, sizeof(ZdBuf));
        sbufs[i].in_off = src_off[i];
        sbufs[i].in_len = source_lens[i];
        sbufs[i].wrap = (uint32_t)wrap;
    }
    DevBuf d_sbufs, d_sres, d_pieces;
    ZlibReturn rc = Z_OK;
    if (!d_sbufs.ensure(sizeof(ZdBuf) * count) || !d_sres.ensure(sizeof(ZdResult) * count))
        rc = Z_MEM_ERROR;
    if (rc == Z_OK && hipMemcpy(d_sbufs.p, sbufs.data(), sizeof(ZdBuf) * count, hipMemcpyHostToDevice) != hipSuccess)
        rc = Z_STREAM_ERROR;

    HipSecRunner runner;
    runner.d_src = d_src;
    runner.src_off = src_off;
    runner.level = level;
    runner.mem_level = mem_level;
    runner.wbits = wbits;
    runner.strategy = strategy;
    runner.streams = &streams;
    {
        /* a run is parsed again from its start whenever a joint turns up behind it; 64 times the input
         * (and 256 MB for small calls) is far beyond what any sensible call needs and bounds the rest */
        uint64_t total = 0;
        for (U32 i = 0; i < count; i++)
            total += source_lens[i];
        runner.budget_bytes = 64ull * total + (256ull << 20);
        if (const char *e = getenv("ZSC_HIP_SECTIONS_BUDGET_MB")) /* (test hook) */
            runner.budget_bytes = (uint64_t)atoll(e) << 20;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    if (rc == Z_OK) {
        const int e = sec_compress(streams, runner);
        if (e != 0)
            rc = runner.error != Z_OK ? runner.error : Z_STREAM_ERROR;
    }
    const auto t_parsed = std::chrono::steady_clock::now();

    /* put the streams together: per round one launch that copies the runs' bytes, one for the
     * headers / markers / trailers */
    std::vector<std::vector<ZdStorePiece>> by_round(runner.rounds.size());
    std::vector<ZdStorePiece> small;
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        if (streams[i].status == SEC_Z_STREAM_ERROR) {
            ZSC_WARN1("zsc_hip: stream %u: the runs of sections do not fit together (a bug).", i);
            rc = Z_STREAM_ERROR;
            break;
        }
        for (const SecPiece &sp : streams[i].pieces) {
            if (sp.dst >= streams[i].delivered)
                continue; /* behind what the caller gets (dest too small) */
            ZdStorePiece pc;
            memset(&pc, 0, sizeof pc);
            pc.dst_off = dst_off[i] + sp.dst;
            pc.len = std::min(sp.len, streams[i].delivered - sp.dst);
            pc.buf = i;
            if (sp.kind == SEC_PIECE_RUN || sp.kind == SEC_PIECE_TAIL) {
                pc.kind = sp.kind == SEC_PIECE_RUN ? 6u : 7u;
                pc.src_off = runner.rounds[sp.round]->out_off[sp.job] + sp.src;
                pc.arg = sp.mask;
                by_round[sp.round].push_back(pc);
                continue;
            }
            /* a header / marker / trailer: k_store writes the first pc.len bytes of it */
            if (sp.kind == SEC_PIECE_HEADER) {
                pc.kind = wrap == 2 && gzip_header_len ? 4u : 2u; /* 4: zsc_api.c writes the caller's header */
                pc.arg = wrap == 1 ? zh : xfl;
            } else {
                pc.kind = sp.kind == SEC_PIECE_MARKER ? 5u : 3u;
            }
            small.push_back(pc);
        }
    }
    size_t most = small.size();
    for (const std::vector<ZdStorePiece> &v : by_round)
        most = std::max(most, v.size());
    if (rc == Z_OK && !d_pieces.ensure(sizeof(ZdStorePiece) * std::max<size_t>(1, most)))
        rc = Z_MEM_ERROR;
    if (rc == Z_OK)
        hipLaunchKernelGGL(k_checksum, dim3(count), dim3(64), 0, nullptr, d_src,
                           (const ZdBuf *)d_sbufs.p, (ZdResult *)d_sres.p, count);
    for (size_t r = 0; r <= by_round.size() && rc == Z_OK; r++) {
        const std::vector<ZdStorePiece> &v = r < by_round.size() ? by_round[r] : small;
        if (v.empty())
            continue;
        if (hipMemcpy(d_pieces.p, v.data(), sizeof(ZdStorePiece) * v.size(), hipMemcpyHostToDevice) != hipSuccess) {
            rc = Z_STREAM_ERROR;
            break;
        }
        const uint8_t *from = r < by_round.size() ? (const uint8_t *)runner.rounds[r]->d_out.p : d_src;
        hipLaunchKernelGGL(k_store, dim3((uint32_t)v.size()), dim3(256), 0, nullptr, from, d_dst,
                           (const ZdStorePiece *)d_pieces.p, (const ZdBuf *)d_sbufs.p,
                           (const ZdResult *)d_sres.p, (uint32_t)v.size());
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess)
            rc = Z_STREAM_ERROR; /* d_pieces is reused by the next launch */
    }
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        dest_lens[i] = streams[i].delivered;
        if (statuses)
            statuses[i] = (I32)streams[i].status;
    }
    if (getenv("ZSC_HIP_SECTIONS_LOG")) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "zsc_hip sections: %u streams, rounds + simulation %.2f ms, putting together %.2f ms; %u rounds, "
                        "%llu input bytes parsed, compressed bytes of rounds peak held %llu\n", count,
                std::chrono::duration<double, std::milli>(t_parsed - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_parsed).count(), (unsigned)runner.rounds.size(),
                (unsigned long long)runner.parsed_bytes, (unsigned long long)runner.held_peak);
    }
    d_sbufs.release();
    d_sres.release();
    d_pieces.release();
    return rc;
}

extern "C" ZlibReturn zsc_hip_compress_sections_device(U32 count, const void *d_input,
                                                       const uint64_t *in_offsets, const U32 *source_lens,
                                                       const U32 *max_block_lens, void *d_output,
                                                       const uint64_t *out_offsets, const U32 *out_caps,
                                                       U32 *dest_lens, I32 *statuses, I32 level,
                                                       I32 window_bits, I32 mem_level, ZlibStrategy strategy)
{
    DeviceScope scope;
    ZSC_ASSERT(d_input != Z_NULL);
    ZSC_ASSERT(in_offsets != Z_NULL);
    ZSC_ASSERT(source_lens != Z_NULL);
    ZSC_ASSERT(max_block_lens != Z_NULL);
    ZSC_ASSERT(d_output != Z_NULL);
    ZSC_ASSERT(out_offsets != Z_NULL);
    ZSC_ASSERT(out_caps != Z_NULL);
    ZSC_ASSERT(dest_lens != Z_NULL);
    if (count == 0)
        return Z_OK;
    return sections_on_device(count, (const uint8_t *)d_input, in_offsets, source_lens, max_block_lens,
                              (uint8_t *)d_output, out_offsets, out_caps, dest_lens, statuses, level,
                              window_bits, mem_level, strategy, 0);
}

extern "C" ZlibReturn zsc_hip_compress_sections_batch(U32 count, const U8 *const *sources,
                                                      const U32 *source_lens, const U32 *max_block_lens,
                                                      U8 *const *dests, U32 *dest_lens, I32 *statuses,
                                                      I32 level, I32 window_bits, I32 mem_level,
                                                      ZlibStrategy strategy, U32 gzip_header_len)
{
    DeviceScope scope;
    ZSC_ASSERT(sources != Z_NULL);
    ZSC_ASSERT(source_lens != Z_NULL);
    ZSC_ASSERT(max_block_lens != Z_NULL);
    ZSC_ASSERT(dests != Z_NULL);
    ZSC_ASSERT(dest_lens != Z_NULL);
    if (count == 0)
        return Z_OK;
    if (zsc_hip_init(-1) != Z_OK)
        return Z_STREAM_ERROR;
    std::vector<uint64_t> src_off(count), dst_off(count);
    std::vector<U32> caps(count), got(count);
    uint64_t in_bytes = 0, out_bytes = 0;
    for (U32 i = 0; i < count; i++) {
        ZSC_ASSERT(sources[i] != Z_NULL);
        ZSC_ASSERT(dests[i] != Z_NULL);
        src_off[i] = in_bytes;
        dst_off[i] = out_bytes;
        caps[i] = dest_lens[i];
        in_bytes += ((uint64_t)source_lens[i] + 15u) & ~15ull;
        /* a stream never grows past its bound by much, whatever dest the caller has */
        U32 bound = 0;
        if (zsc_compress_get_max_output_size2(source_lens[i], max_block_lens[i], level, window_bits,
                                              mem_level, &bound) != Z_OK)
            bound = dest_lens[i];
        const uint64_t room = std::min<uint64_t>(dest_lens[i], (uint64_t)bound + gzip_header_len +
                                                                   source_lens[i] / max_block_lens[i] * 8ull + 4096u);
        out_bytes += (room + 15u) & ~15ull;
    }
    DevBuf d_src, d_dst;
    ZlibReturn rc = Z_OK;
    if (!d_src.ensure(in_bytes + 64) || !d_dst.ensure(out_bytes + 64))
        rc = Z_MEM_ERROR;
    for (U32 i = 0; i < count && rc == Z_OK; i++)
        if (source_lens[i] && hipMemcpy((uint8_t *)d_src.p + src_off[i], sources[i], source_lens[i],
                                        hipMemcpyHostToDevice) != hipSuccess)
            rc = Z_STREAM_ERROR;
    if (rc == Z_OK)
        rc = sections_on_device(count, (const uint8_t *)d_src.p, src_off.data(), source_lens, max_block_lens,
                                (uint8_t *)d_dst.p, dst_off.data(), caps.data(), got.data(), statuses, level,
                                window_bits, mem_level, strategy, gzip_header_len);
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        if (got[i] && hipMemcpy(dests[i], (uint8_t *)d_dst.p + dst_off[i], got[i], hipMemcpyDeviceToHost) != hipSuccess)
            rc = Z_STREAM_ERROR;
        dest_lens[i] = got[i];
    }
    d_src.release();
    d_dst.release();
    return rc;
}

/* host-pointer batch: stage through one pair of device buffers */
extern "C" ZlibReturn zsc_hip_compress_batch(U32 count, const U8 *const *sources,
                                             const U32 *source_lens, U8 *const *dests,
                                             U32 *dest_lens, I32 *statuses, I32 level,
                                             I32 window_bits, I32 mem_level,
                                             ZlibStrategy strategy)
{
    DeviceScope scope;
    ZSC_ASSERT(sources != Z_NULL);
    ZSC_ASSERT(source_lens != Z_NULL);
    ZSC_ASSERT(dests != Z_NULL);
    ZSC_ASSERT(dest_lens != Z_NULL);
    if (count == 0)
        return Z_OK;
    if (level == Z_NO_COMPRESSION) {
        /* level 0: stored blocks; max_block_len = source_len, the smallest a single section allows */
        std::vector<U32> mbl(count);
        for (U32 i = 0; i < count; i++)
            mbl[i] = source_lens[i] ? source_lens[i] : 1u;
        return zsc_hip_store_batch(count, sources, source_lens, mbl.data(), dests, dest_lens, statuses,
                                   window_bits, mem_level, 0);
    }
    std::vector<uint64_t> in_off(count), out_off(count);
    std::vector<U32> caps(count), lens(count);
    std::vector<I32> stat(count);
    uint64_t in_bytes = 0, out_bytes = 0;
    ZlibReturn rc = zsc_hip_deflate_plan_layout(count, source_lens, level, window_bits, mem_level,
                                                in_off.data(), out_off.data(), caps.data(),
                                                &in_bytes, &out_bytes);
    if (rc != Z_OK)
        return rc;
    zsc_hip_deflate_plan *pl = nullptr;
    rc = zsc_hip_deflate_plan_create(&pl, count, source_lens, in_off.data(), out_off.data(),
                                     caps.data(), level, window_bits, mem_level, strategy);
    if (rc != Z_OK)
        return rc;
    DevBuf d_in, d_out;
    if (!d_in.ensure(in_bytes) || !d_out.ensure(out_bytes)) {
        d_in.release();
        d_out.release();
        zsc_hip_deflate_plan_destroy(pl);
        return Z_MEM_ERROR;
    }
    rc = Z_OK;
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        ZSC_ASSERT(sources[i] != Z_NULL);
        if (source_lens[i] &&
            hipMemcpy((uint8_t *)d_in.p + in_off[i], sources[i], source_lens[i],
                      hipMemcpyHostToDevice) != hipSuccess)
            rc = Z_STREAM_ERROR;
    }
    if (rc == Z_OK)
        rc = zsc_hip_deflate_plan_run(pl, d_in.p, d_out.p, nullptr);
    if (rc == Z_OK)
        rc = zsc_hip_deflate_plan_results(pl, lens.data(), stat.data());
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        ZSC_ASSERT(dests[i] != Z_NULL);
        const U32 cap = dest_lens[i];
        I32 s = stat[i];
        U32 give = lens[i];
        if (s == Z_OK && give > cap) {
            /* the reference hands out dest in slices and fails once it is used up:
             * the caller keeps the prefix that fitted (src/zsc_compress.c:126-140) */
            give = cap;
            s = Z_BUF_ERROR;
        } else if (s != Z_OK) {
            give = 0;
        }
        if (give && hipMemcpy(dests[i], (uint8_t *)d_out.p + out_off[i], give,
                              hipMemcpyDeviceToHost) != hipSuccess)
            rc = Z_STREAM_ERROR;
        dest_lens[i] = give;
        if (statuses)
            statuses[i] = s;
    }
    d_in.release();
    d_out.release();
    zsc_hip_deflate_plan_destroy(pl);
    return rc;
}

/* wavefronts of k_inflate: enough to fill every CU at its occupancy, never more than the streams need */
static uint32_t inflate_grid(uint32_t count)
{
    const uint32_t need = (count + INF_PER_WAVE - 1) / INF_PER_WAVE;
    const uint32_t fill = (uint32_t)g_cus * 4u * INF_WAVES_EU;
    return std::max(1u, std::min(need, fill));
}

struct zsc_hip_inflate_plan {
    uint32_t count = 0;
    int32_t window_bits = 15;
    DevBuf d_items, d_order, d_res, d_resume, d_pending;
    const void *last_src = nullptr;
    void *last_dst = nullptr;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

extern "C" ZlibReturn zsc_hip_inflate_plan_create_ordered(zsc_hip_inflate_plan **plan_out, U32 count,
                                                  const U32 *source_lens,
                                                  const uint64_t *src_offsets,
                                                  const U32 *dest_caps,
                                                  const uint64_t *dst_offsets, I32 window_bits,
                                                       const U32 *decode_order)
{
    DeviceScope scope;
    ZSC_ASSERT(plan_out != Z_NULL);
    *plan_out = nullptr;
    if (zsc_hip_init(-1) != Z_OK)
        return Z_STREAM_ERROR;
    auto *pl = new zsc_hip_inflate_plan();
    pl->count = count;
    pl->window_bits = window_bits;
    std::vector<ZdInfItem> items(count);
    std::vector<uint32_t> order(count);
    for (U32 i = 0; i < count; i++) {
        items[i].src_off = src_offsets[i];
        items[i].dst_off = dst_offsets[i];
        items[i].src_len = source_lens[i];
        items[i].dst_cap = dest_caps[i];
        order[i] = i;
        if ((src_offsets[i] | dst_offsets[i]) & 15u) {
            /* (the decoder reads a stream's last, partial dword whole: it must not straddle a page) */
            ZSC_WARN1("zsc_hip: stream %u is not 16-byte aligned in the batch.", i);
            delete pl;
            return Z_STREAM_ERROR;
        }
    }
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return items[a].dst_cap > items[b].dst_cap; });
    if (decode_order) {
        /* the caller's order: a permutation of the streams (checked), e.g. to keep the replicas of one
         * stream of a benchmark batch from sharing wavefronts (bench.py) */
        std::vector<uint8_t> seen(count, 0);
        for (uint32_t k = 0; k < count; k++) {
            if (decode_order[k] >= count || seen[decode_order[k]]) {
                ZSC_WARN1("zsc_hip: decode_order is not a permutation (entry %u).", k);
                delete pl;
                return Z_STREAM_ERROR;
            }
            seen[decode_order[k]] = 1;
            order[k] = decode_order[k];
        }
    }
    bool ok = pl->d_items.ensure(sizeof(ZdInfItem) * std::max(1u, count)) &&
              pl->d_order.ensure(4ull * std::max(1u, count)) &&
              pl->d_res.ensure(sizeof(InfResult) * std::max(1u, count)) &&
              pl->d_resume.ensure(sizeof(InfResume) * std::max(1u, count)) && pl->d_pending.ensure(8);
    if (ok && count) {
        ok = hipMemcpy(pl->d_items.p, items.data(), sizeof(ZdInfItem) * count,
                       hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(pl->d_order.p, order.data(), 4ull * count, hipMemcpyHostToDevice) ==
                 hipSuccess;
    }
    if (!ok) {
        pl->d_items.release();
        pl->d_order.release();
        pl->d_res.release();
        pl->d_resume.release();
        pl->d_pending.release();
        delete pl;
        return Z_MEM_ERROR;
    }
    (void)hipEventCreate(&pl->ev0);
    (void)hipEventCreate(&pl->ev1);
    *plan_out = pl;
    return Z_OK;
}

extern "C" ZlibReturn zsc_hip_inflate_plan_create(zsc_hip_inflate_plan **plan_out, U32 count,
                                                  const U32 *source_lens,
                                                  const uint64_t *src_offsets,
                                                  const U32 *dest_caps,
                                                  const uint64_t *dst_offsets, I32 window_bits)
{
    return zsc_hip_inflate_plan_create_ordered(plan_out, count, source_lens, src_offsets, dest_caps, dst_offsets,
                                               window_bits, Z_NULL);
}

extern "C" ZlibReturn zsc_hip_inflate_plan_run(zsc_hip_inflate_plan *pl, const void *d_src,
                                               void *d_dst, void *hip_stream)
{
    DeviceScope scope;
    ZSC_ASSERT(pl != Z_NULL);
    hipStream_t st = (hipStream_t)hip_stream;
    pl->last_stream = st;
    if (pl->count == 0)
        return Z_OK;
    (void)hipGetLastError(); /* (see zsc_hip_deflate_plan_run) */
    pl->last_src = d_src;
    pl->last_dst = d_dst;
    HIP_TRY(hipMemsetAsync(pl->d_resume.p, 0, sizeof(InfResume) * pl->count, st), return Z_STREAM_ERROR);
    HIP_TRY(hipMemsetAsync(pl->d_pending.p, 0, 8, st), return Z_STREAM_ERROR); /* [0] streams to relaunch, [1] the queue */
    (void)hipEventRecord(pl->ev0, st);
    hipLaunchKernelGGL(k_inflate, dim3(inflate_grid(pl->count)), dim3(64), 0, st, (const uint8_t *)d_src,
                       (uint8_t *)d_dst, (const ZdInfItem *)pl->d_items.p,
                       (const uint32_t *)pl->d_order.p, (InfResult *)pl->d_res.p,
                       (InfResume *)pl->d_resume.p, (uint32_t *)pl->d_pending.p, pl->window_bits,
                       pl->count);
    (void)hipEventRecord(pl->ev1, st);
    pl->timed = true;
    HIP_TRY(hipGetLastError(), return Z_STREAM_ERROR);
    return Z_OK;
}

extern "C" ZlibReturn zsc_hip_inflate_plan_results(zsc_hip_inflate_plan *pl, U32 *dest_lens,
                                                   U32 *consumed, I32 *statuses, float *kernel_ms)
{
    DeviceScope scope;
    ZSC_ASSERT(pl != Z_NULL);
    HIP_TRY(hipStreamSynchronize(pl->last_stream), return Z_STREAM_ERROR);
    /* streams that hit a data error and found a flush marker behind it (inflateSync) are
     * inflated again from there, as zsc_uncompress's loop does (src/zsc_uncompr.c:104-125);
     * every round consumes at least the marker, so this ends */
    for (uint32_t round = 0; pl->count && pl->last_dst; round++) {
        uint32_t pending = 0;
        HIP_TRY(hipMemcpy(&pending, pl->d_pending.p, 4, hipMemcpyDeviceToHost), return Z_STREAM_ERROR);
        if (pending == 0 || round > (1u << 30))
            break;
        HIP_TRY(hipMemsetAsync(pl->d_pending.p, 0, 8, pl->last_stream), return Z_STREAM_ERROR);
        hipLaunchKernelGGL(k_inflate, dim3(inflate_grid(pl->count)), dim3(64), 0, pl->last_stream,
                           (const uint8_t *)pl->last_src, (uint8_t *)pl->last_dst,
                           (const ZdInfItem *)pl->d_items.p, (const uint32_t *)pl->d_order.p,
                           (InfResult *)pl->d_res.p, (InfResume *)pl->d_resume.p,
                           (uint32_t *)pl->d_pending.p, pl->window_bits, pl->count);
        HIP_TRY(hipStreamSynchronize(pl->last_stream), return Z_STREAM_ERROR);
    }
    std::vector<InfResult> res(pl->count);
    if (pl->count)
        HIP_TRY(hipMemcpy(res.data(), pl->d_res.p, sizeof(InfResult) * pl->count,
                          hipMemcpyDeviceToHost),
                return Z_STREAM_ERROR);
    for (uint32_t i = 0; i < pl->count; i++) {
        if (dest_lens)
            dest_lens[i] = res[i].out_len;
        if (consumed)
            consumed[i] = res[i].consumed;
        if (statuses)
            statuses[i] = res[i].status;
        if (res[i].status == Z_DATA_ERROR && getenv("ZSC_HIP_DEBUG"))
            fprintf(stderr, "zsc_hip: stream %u rejected at inflate.h:%u\n", i, res[i].pad);
    }
    if (kernel_ms) {
        *kernel_ms = 0.f;
        if (pl->timed)
            (void)hipEventElapsedTime(kernel_ms, pl->ev0, pl->ev1);
    }
    return Z_OK;
}

extern "C" void zsc_hip_inflate_plan_destroy(zsc_hip_inflate_plan *pl)
{
    DeviceScope scope;
    if (!pl)
        return;
    (void)hipStreamSynchronize(pl->last_stream);
    pl->d_items.release();
    pl->d_order.release();
    pl->d_res.release();
    pl->d_resume.release();
    pl->d_pending.release();
    if (pl->ev0)
        (void)hipEventDestroy(pl->ev0);
    if (pl->ev1)
        (void)hipEventDestroy(pl->ev1);
    delete pl;
}

/* host-pointer batch: stage through one pair of device buffers */
extern "C" ZlibReturn zsc_hip_uncompress_batch(U32 count, const U8 *const *sources,
                                               U32 *source_lens, U8 *const *dests,
                                               U32 *dest_lens, I32 *statuses, I32 window_bits)
{
    DeviceScope scope;
    ZSC_ASSERT(sources != Z_NULL);
    ZSC_ASSERT(source_lens != Z_NULL);
    ZSC_ASSERT(dests != Z_NULL);
    ZSC_ASSERT(dest_lens != Z_NULL);
    if (count == 0)
        return Z_OK;
    if (zsc_hip_init(-1) != Z_OK)
        return Z_STREAM_ERROR;
    std::vector<uint64_t> so(count), dof(count);
    uint64_t sb = 0, db = 0;
    for (U32 i = 0; i < count; i++) {
        so[i] = sb;
        dof[i] = db;
        sb += ((uint64_t)source_lens[i] + 64u + 15u) & ~15ull;
        db += ((uint64_t)dest_lens[i] + 64u + 15u) & ~15ull;
    }
    zsc_hip_inflate_plan *pl = nullptr;
    ZlibReturn rc = zsc_hip_inflate_plan_create(&pl, count, source_lens, so.data(), dest_lens,
                                                dof.data(), window_bits);
    if (rc != Z_OK)
        return rc;
    DevBuf d_src, d_dst;
    if (!d_src.ensure(sb + 64) || !d_dst.ensure(db + 64)) {
        d_src.release();
        d_dst.release();
        zsc_hip_inflate_plan_destroy(pl);
        return Z_MEM_ERROR;
    }
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        ZSC_ASSERT(sources[i] != Z_NULL);
        if (source_lens[i] && hipMemcpy((uint8_t *)d_src.p + so[i], sources[i], source_lens[i],
                                        hipMemcpyHostToDevice) != hipSuccess)
            rc = Z_STREAM_ERROR;
    }
    std::vector<U32> outl(count), used(count);
    std::vector<I32> stat(count);
    if (rc == Z_OK)
        rc = zsc_hip_inflate_plan_run(pl, d_src.p, d_dst.p, nullptr);
    if (rc == Z_OK)
        rc = zsc_hip_inflate_plan_results(pl, outl.data(), used.data(), stat.data(), nullptr);
    for (U32 i = 0; i < count && rc == Z_OK; i++) {
        ZSC_ASSERT(dests[i] != Z_NULL);
        if (outl[i] && hipMemcpy(dests[i], (uint8_t *)d_dst.p + dof[i], outl[i],
                                 hipMemcpyDeviceToHost) != hipSuccess)
            rc = Z_STREAM_ERROR;
        dest_lens[i] = outl[i];
        source_lens[i] = used[i];
        if (statuses)
            statuses[i] = stat[i];
    }
    d_src.release();
    d_dst.release();
    zsc_hip_inflate_plan_destroy(pl);
    return rc;
}
