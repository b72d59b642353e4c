/*
 * hash_sort.h -- kernel 1: per-tile hash-chain construction by sorting.
 *
 * Replaces the reference's INSERT_STRING / head[] / prev[] machinery
 * (src/deflate.c:174-189, slide_hash :204-231).  For compression levels 4-9
 * every position is inserted into the chain of its 3-byte hash
 * (src/deflate.c:2018,2069-2075), so "the chain of p" is simply: all earlier
 * positions with the same hash, newest first.  Sorting a tile's positions by
 * (hash, position) makes that chain a CONTIGUOUS, descending run that a
 * wavefront can fetch 64 candidates at a time with one coalesced load, instead
 * of chasing prev[] links one dependent load at a time.
 *
 * One workgroup of HS_WAVES wavefronts per tile of 32 768 positions, and the whole
 * sort happens in LDS: the tile's bytes are read from HBM once (16-byte loads), the
 * sorted order, the rank of every position and the bucket directory are written once,
 * coalesced.  (Round 1 scattered 4-byte entries through global memory in both passes;
 * on gfx9 a load waits for the stores issued before it -- one vmcnt for both -- so
 * every 64-element step paid a store and a load latency back to back.)
 *
 * Stable LSD radix sort in two passes over the 15-bit hash (8 low bits, 7 high bits)
 * of 16-bit positions held in LDS (`ord`); the hash is recomputed from the tile's
 * bytes (also in LDS) wherever it is needed, so an entry stays 2 bytes:
 *   0  clear counters, tile bytes -> LDS
 *   1  per-wave-slice histogram of the low digit
 *   2,3  exclusive scan (digit-major, wave-minor), all waves
 *   4  stable scatter ord[dst] = position; rank inside a 64-element step comes from
 *      the lanes' digit match (HS_MATCH: an order-independent OR into LDS), not from
 *      fetch-and-add; the high digit is counted per DESTINATION slice on the way
 *   5,6  scan
 *   7  second pass: every wave reads its slice of ord and computes destinations
 *      into registers...
 *   8  ...and writes them after the barrier (the permutation is done in place)
 *   9  sorted[] / sorted16[] and the bucket directory dir[h] = first sorted index
 *      with hash >= h, coalesced; the slice's positions stay in registers...
 *   10 ...ord[position] = sorted index
 *   11 rank[] <- ord, coalesced
 * LDS: 32 KiB bytes + 64 KiB ord + 24 KiB counters + 32 KiB match tables = one workgroup per CU.
 *
 * Bytes per input byte (TILE positions): read 1 + sorted 4w (+2w sorted16) + rank 2w + dir 2w.
 */
#ifndef ZSC_HASH_SORT_H
#define ZSC_HASH_SORT_H

#include "wave.h"
#include "zsc_dev.h"

#define HS_WAVES 16
#define HS_STEPS (ZD_TILE / (HS_WAVES * WAVE)) /* 64-element steps of one wave's slice, at most */

typedef struct {
    __attribute__((aligned(16))) uint8_t in[ZD_TILE + 16];  /* the tile's bytes (+2 for the last strings, + slack for lds_u32) */
    uint16_t ord[ZD_TILE];     /* positions in pass order; at the end: rank by position */
    uint32_t cnt0[256 * HS_WAVES];
    uint32_t cnt1[128 * HS_WAVES];
    uint32_t tot[HS_WAVES];
    uint32_t match[HS_WAVES][2 * 256]; /* per wave and digit: the lanes of the current step that hold it */
} HsLds;

/* what a wave keeps in registers from one phase to the next */
typedef struct {
    LANEARR(uint32_t, r, HS_STEPS);
} HsRegs;

typedef struct {
    const uint8_t *in;  /* the tile's buffer (16-byte aligned) */
    uint32_t n;         /* buffer length */
    uint32_t start;     /* absolute position of the tile's first byte */
    uint32_t m;         /* positions of this tile that own a 3-byte string (pos <= n-3) */
    uint32_t *sorted;   /* TILE entries */
    uint16_t *sorted16; /* the same order, positions only (what the lane-per-segment parser reads); may be null */
    uint16_t *rank;     /* per position of the buffer */
    uint16_t *dir;      /* DIR_STRIDE entries: first sorted index of every bucket */
    const uint16_t *dir_prev; /* directory of the previous tile of the same buffer, or null */
    uint16_t *hib;      /* per position of the buffer: last sorted index of its bucket in the previous tile */
    uint32_t *cnt;      /* per position of the buffer: earlier members of its bucket in its own tile (low
                           half) and members of the bucket in the previous tile (high half) */
    uint64_t *meta;     /* per position of the buffer: rank | hib << 16 | cnt << 32 in one record; may be null */
} HsTile;

/* UPDATE_HASH over three bytes, reference src/deflate.c:174-175 */
DEV uint32_t hs_hash3(const uint8_t *in, uint32_t pos, uint32_t n)
{
    uint32_t b0, b1, b2;
    if (pos + 4 <= n) {
        uint32_t w = ld_u32(in + pos);
        b0 = w & 0xff;
        b1 = (w >> 8) & 0xff;
        b2 = (w >> 16) & 0xff;
    } else {
        b0 = in[pos];
        b1 = in[pos + 1];
        b2 = in[pos + 2];
    }
    return ((b0 << 10) ^ (b1 << 5) ^ b2) & ZD_HASH_MASK;
}

/* the same from the tile's bytes in LDS (pos relative to the tile) */
DEV uint32_t hs_hash_lds(const HsLds *lds, uint32_t pos)
{
    const uint32_t w = lds_u32(lds->in, pos);
    return (((w & 0xff) << 10) ^ (((w >> 8) & 0xff) << 5) ^ ((w >> 16) & 0xff)) & ZD_HASH_MASK;
}

DEV uint32_t hs_slice(uint32_t m)
{
    uint32_t per = (m + HS_WAVES * WAVE - 1) / (HS_WAVES * WAVE);
    return per * WAVE;
}

/* x / per for x < 2^16 as a multiply-high (per = steps of a slice, 1..HS_STEPS) */
DEV uint32_t hs_div(uint32_t x, uint32_t per, uint32_t magic)
{
    const uint32_t q = (uint32_t)(((uint64_t)x * magic) >> 32);
    return per == 1 ? x : q;
}

DEV void hs_load_tile(const HsTile &t, HsLds *lds, int w)
{
    const uint32_t avail = t.n > t.start ? (t.n - t.start < ZD_TILE + 2u ? t.n - t.start : ZD_TILE + 2u) : 0u;
    const uint8_t *src = t.in + t.start;
    const uint32_t nvec = avail / 16u;
    for (uint32_t v = (uint32_t)w * WAVE; v < nvec; v += HS_WAVES * WAVE) {
        FOR_LANES
        {
            const uint32_t k = v + (uint32_t)LANE;
            if (k < nvec)
                COPY16(lds->in + 16u * k, src + 16u * k);
        }
    }
    if (w == 0) {
        FOR_LANES
        {
            const uint32_t k = nvec * 16u + (uint32_t)LANE;
            if (LANE < 16 && k < avail)
                lds->in[k] = src[k];
        }
    }
}

DEV void hs_count(const HsTile &t, HsLds *lds, int w)
{
    const uint32_t slice = hs_slice(t.m);
    const uint32_t lo = (uint32_t)w * slice;
    for (uint32_t s = lo; s < lo + slice && s < t.m; s += WAVE) {
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            if (i < t.m)
                LDS_ADD_U32(&lds->cnt0[(hs_hash_lds(lds, i) & 0xff) * HS_WAVES + (uint32_t)w], 1u);
        }
    }
}

/* exclusive scan of ndig*HS_WAVES counters, digit-major: every wave scans its ndig counters (a),
 * then adds the waves before it (b) */
DEV void hs_scan_a(uint32_t *cnt, uint32_t *tot, int ndig, int w)
{
    uint32_t run = 0;
    for (int s = w * ndig; s < (w + 1) * ndig; s += WAVE) {
        LANEVAR(uint32_t, v);
        LANEVAR(uint32_t, ex);
        FOR_LANES { LV(v) = cnt[s + LANE]; }
        uint32_t sum;
        WAVE_EXSCAN(v, ex, sum);
        FOR_LANES { cnt[s + LANE] = run + LV(ex); }
        run += sum;
    }
    FOR_LANES
    {
        if (LANE == 0)
            tot[w] = run;
    }
}

DEV void hs_scan_b(uint32_t *cnt, const uint32_t *tot, int ndig, int w)
{
    uint32_t before = 0;
    for (int k = 0; k < w; k++)
        before += tot[k];
    before = UNI(before);
    for (int s = w * ndig; s < (w + 1) * ndig; s += WAVE) {
        FOR_LANES { cnt[s + LANE] += before; }
    }
}

/* which live lanes of this step hold the same digit: every lane ORs its bit into its digit's
 * 64-bit word of the wave's match table, reads the word back, and the first lane of each group
 * clears it again (LDS operations of one wave execute in order).  3 LDS operations instead of
 * the ballot multi-split's 6 vector instructions per digit bit -- the sort was VALU-bound on it.
 * before = lanes of my digit below me, all = lanes of my digit. */
#define HS_MATCH(tab, dig, ok, before, all)                                       \
    do {                                                                          \
        FOR_LANES                                                                 \
        {                                                                         \
            if (LV(ok))                                                           \
                LDS_OR_U32(&(tab)[2u * LV(dig) + ((uint32_t)LANE >> 5)], 1u << ((uint32_t)LANE & 31u)); \
        }                                                                         \
        WAVE_SYNC();                                                              \
        LANEVAR(uint64_t, peers_);                                                \
        FOR_LANES                                                                 \
        {                                                                         \
            LV(peers_) = 0;                                                       \
            if (LV(ok))                                                           \
                LV(peers_) = (uint64_t)(tab)[2u * LV(dig)] | ((uint64_t)(tab)[2u * LV(dig) + 1u] << 32); \
            LV(before) = (uint32_t)POPC64(LV(peers_) & ((1ull << LANE) - 1ull));  \
            LV(all) = (uint32_t)POPC64(LV(peers_));                               \
        }                                                                         \
        WAVE_SYNC();                                                              \
        FOR_LANES                                                                 \
        {                                                                         \
            if (LV(ok) && LV(before) == 0) {                                      \
                (tab)[2u * LV(dig)] = 0;                                          \
                (tab)[2u * LV(dig) + 1u] = 0;                                     \
            }                                                                     \
        }                                                                         \
    } while (0)

/* pass 1: by the low 8 bits, positions in order -> ord */
DEV void hs_scatter_low(const HsTile &t, HsLds *lds, int w)
{
    const uint32_t slice = hs_slice(t.m);
    const uint32_t lo = (uint32_t)w * slice;
    const uint32_t per = slice / WAVE;
    const uint32_t magic = per > 1 ? (uint32_t)(((1ull << 32) + per - 1) / per) : 0u;
    for (uint32_t s = lo; s < lo + slice && s < t.m; s += WAVE) {
        LANEVAR(uint32_t, h);
        LANEVAR(uint32_t, dig);
        LANEVAR(int, ok);
        LANEVAR(uint32_t, before);
        LANEVAR(uint32_t, all);
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            LV(ok) = i < t.m;
            LV(h) = LV(ok) ? hs_hash_lds(lds, i) : 0u;
            LV(dig) = LV(h) & 0xff;
        }
        HS_MATCH(lds->match[w], dig, ok, before, all);
        LANEVAR(uint32_t, dst);
        FOR_LANES
        {
            if (LV(ok))
                LV(dst) = lds->cnt0[LV(dig) * HS_WAVES + (uint32_t)w] + LV(before);
        }
        FOR_LANES
        {
            if (LV(ok)) {
                if (LV(before) == 0) /* first lane of its digit group */
                    lds->cnt0[LV(dig) * HS_WAVES + (uint32_t)w] += LV(all);
                lds->ord[LV(dst)] = (uint16_t)(s + (uint32_t)LANE);
                /* the second pass counts its digits per wave slice of ord: where this entry lands
                 * decides the slice, so that count is taken here */
                LDS_ADD_U32(&lds->cnt1[(LV(h) >> 8) * HS_WAVES + hs_div(LV(dst) / WAVE, per, magic)], 1u);
            }
        }
    }
}

/* pass 2, first half: destinations of this wave's slice of ord, into registers */
DEV void hs_scatter_high_plan(const HsTile &t, HsLds *lds, HsRegs *rg, int w)
{
    const uint32_t slice = hs_slice(t.m);
    const uint32_t lo = (uint32_t)w * slice;
    UNROLL_FULL
    for (uint32_t k = 0; k < HS_STEPS; k++) {
        const uint32_t s = lo + k * WAVE;
        LANEVAR(uint32_t, pos);
        LANEVAR(uint32_t, dig);
        LANEVAR(int, ok);
        LANEVAR(uint32_t, before);
        LANEVAR(uint32_t, all);
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            LV(ok) = k * WAVE < slice && i < t.m;
            LV(pos) = LV(ok) ? (uint32_t)lds->ord[i] : 0u;
            LV(dig) = LV(ok) ? hs_hash_lds(lds, LV(pos)) >> 8 : 0u;
        }
        FOR_LANES { LVA(rg->r, k) = 0xffffffffu; }
        if (BALLOT(ok) != 0) { /* (uniform) */
            HS_MATCH(lds->match[w], dig, ok, before, all);
            FOR_LANES
            {
                if (LV(ok)) {
                    const uint32_t dst = lds->cnt1[LV(dig) * HS_WAVES + (uint32_t)w] + LV(before);
                    LVA(rg->r, k) = LV(pos) | (dst << 16);
                }
            }
            FOR_LANES
            {
                if (LV(ok) && LV(before) == 0)
                    lds->cnt1[LV(dig) * HS_WAVES + (uint32_t)w] += LV(all);
            }
        }
    }
}

/* pass 2, second half (every wave has read its slice): the in-place permutation */
DEV void hs_scatter_high_store(HsLds *lds, const HsRegs *rg)
{
    UNROLL_FULL
    for (uint32_t k = 0; k < HS_STEPS; k++) {
        FOR_LANES
        {
            const uint32_t e = LVA(rg->r, k);
            if (e != 0xffffffffu)
                lds->ord[e >> 16] = (uint16_t)(e & 0xffffu);
        }
    }
}

/* ord is sorted by (hash, position): write it out, and the bucket starts -- sorted index i
 * opens every bucket in (h[i-1], h[i]] */
DEV void hs_output(const HsTile &t, HsLds *lds, HsRegs *rg, int w)
{
    const uint32_t slice = hs_slice(t.m);
    const uint32_t lo = (uint32_t)w * slice;
    UNROLL_FULL
    for (uint32_t k = 0; k < HS_STEPS; k++) {
        FOR_LANES
        {
            const uint32_t i = lo + k * WAVE + (uint32_t)LANE;
            uint32_t e = 0xffffffffu;
            if (k * WAVE < slice && i < t.m) {
                const uint32_t pos = lds->ord[i];
                const uint32_t h = hs_hash_lds(lds, pos);
                const int32_t hp = i == 0 ? -1 : (int32_t)hs_hash_lds(lds, lds->ord[i - 1]);
                t.sorted[i] = pos | (h << 16);
                if (t.sorted16)
                    t.sorted16[i] = (uint16_t)pos;
                for (int32_t hh = hp + 1; hh <= (int32_t)h; hh++)
                    t.dir[hh] = (uint16_t)i;
                e = pos | (i << 16);
            }
            LVA(rg->r, k) = e;
        }
    }
    /* buckets past the last occupied one (and the end sentinel) point at m */
    int32_t hl = -1;
    if (t.m != 0)
        hl = (int32_t)UNI(hs_hash_lds(lds, lds->ord[t.m - 1]));
    for (int32_t s = hl + 1 + w * WAVE; s <= 32768; s += HS_WAVES * WAVE) {
        FOR_LANES
        {
            int32_t hh = s + LANE;
            if (hh <= 32768)
                t.dir[hh] = (uint16_t)t.m;
        }
    }
}

DEV void hs_rank_store(HsLds *lds, const HsRegs *rg)
{
    UNROLL_FULL
    for (uint32_t k = 0; k < HS_STEPS; k++) {
        FOR_LANES
        {
            const uint32_t e = LVA(rg->r, k);
            if (e != 0xffffffffu)
                lds->ord[e & 0xffffu] = (uint16_t)(e >> 16);
        }
    }
}

DEV void hs_rank_out(const HsTile &t, const HsLds *lds, int w)
{
    for (uint32_t s = (uint32_t)w * WAVE; s < t.m; s += HS_WAVES * WAVE) {
        FOR_LANES
        {
            const uint32_t i = s + (uint32_t)LANE;
            if (i < t.m)
                t.rank[t.start + i] = lds->ord[i];
        }
    }
}

/* kernel 1b (after every tile of the batch has its directory): for each position of
 * this tile, how long its chain is inside the tile, and where its hash bucket lies in
 * the PREVIOUS tile's sorted array.  With rank[] this makes a position's whole chain
 * addressable up front -- the parser never does a dependent table lookup and never
 * has to look at an entry to find the end of a bucket. */
#define HS_LINK_BATCH 4 /* steps whose loads are all issued before the first store (a gfx9 load waits for earlier stores) */
DEV void hs_link_prev(const HsTile &t, int w)
{
    for (uint32_t s = (uint32_t)w * WAVE; s < t.m; s += HS_LINK_BATCH * HS_WAVES * WAVE) {
        LANEARR(uint32_t, h, HS_LINK_BATCH);
        LANEARR(uint32_t, rk, HS_LINK_BATCH);
        LANEARR(uint32_t, d0, HS_LINK_BATCH);
        LANEARR(uint32_t, plo, HS_LINK_BATCH);
        LANEARR(uint32_t, pend, HS_LINK_BATCH);
        UNROLL_FULL
        for (uint32_t k = 0; k < HS_LINK_BATCH; k++) {
            FOR_LANES
            {
                const uint32_t i = s + k * HS_WAVES * WAVE + (uint32_t)LANE;
                LVA(h, k) = 0;
                LVA(rk, k) = 0;
                if (i < t.m) {
                    LVA(h, k) = hs_hash3(t.in, t.start + i, t.n);
                    LVA(rk, k) = (uint32_t)t.rank[t.start + i];
                }
            }
        }
        UNROLL_FULL
        for (uint32_t k = 0; k < HS_LINK_BATCH; k++) {
            FOR_LANES
            {
                const uint32_t i = s + k * HS_WAVES * WAVE + (uint32_t)LANE;
                LVA(d0, k) = 0;
                LVA(plo, k) = 0;
                LVA(pend, k) = 0;
                if (i < t.m) {
                    LVA(d0, k) = t.dir[LVA(h, k)];
                    if (t.dir_prev) {
                        /* dir_prev[h] and dir_prev[h + 1] in one (2-byte aligned) load: the kernel is
                         * bound by the rate of its scattered reads */
                        const uint32_t two = ld_u32((const uint8_t *)(t.dir_prev + LVA(h, k)));
                        LVA(plo, k) = two & 0xffffu;
                        LVA(pend, k) = two >> 16;
                    }
                }
            }
        }
        UNROLL_FULL
        for (uint32_t k = 0; k < HS_LINK_BATCH; k++) {
            FOR_LANES
            {
                const uint32_t i = s + k * HS_WAVES * WAVE + (uint32_t)LANE;
                if (i < t.m) {
                    uint32_t c = LVA(rk, k) - LVA(d0, k), hb = 0xffffu;
                    if (t.dir_prev) {
                        hb = (LVA(pend, k) - 1u) & 0xffffu; /* 0xffff: bucket empty from the start */
                        t.hib[t.start + i] = (uint16_t)hb;
                        c |= (LVA(pend, k) - LVA(plo, k)) << 16;
                    }
                    t.cnt[t.start + i] = c;
                    if (t.meta)
                        t.meta[t.start + i] = (uint64_t)(LVA(rk, k) | (hb << 16)) | ((uint64_t)c << 32);
                }
            }
        }
    }
}

/* one phase of the tile sort, executed by wave `w` of the tile's workgroup;
 * the caller puts a workgroup barrier between phases */
DEV void hash_sort_phase(const HsTile &t, HsLds *lds, HsRegs *rg, int w, int phase)
{
    switch (phase) {
    case 0:
        for (int i = w * WAVE; i < 256 * HS_WAVES; i += HS_WAVES * WAVE) {
            FOR_LANES { lds->cnt0[i + LANE] = 0; }
        }
        for (int i = w * WAVE; i < 128 * HS_WAVES; i += HS_WAVES * WAVE) {
            FOR_LANES { lds->cnt1[i + LANE] = 0; }
        }
        for (int i = 0; i < 2 * 256; i += WAVE) {
            FOR_LANES { lds->match[w][i + LANE] = 0; }
        }
        hs_load_tile(t, lds, w);
        break;
    case 1:
        hs_count(t, lds, w);
        break;
    case 2:
        hs_scan_a(lds->cnt0, lds->tot, 256, w);
        break;
    case 3:
        hs_scan_b(lds->cnt0, lds->tot, 256, w);
        break;
    case 4:
        hs_scatter_low(t, lds, w);
        break;
    case 5:
        hs_scan_a(lds->cnt1, lds->tot, 128, w);
        break;
    case 6:
        hs_scan_b(lds->cnt1, lds->tot, 128, w);
        break;
    case 7:
        hs_scatter_high_plan(t, lds, rg, w);
        break;
    case 8:
        hs_scatter_high_store(lds, rg);
        break;
    case 9:
        hs_output(t, lds, rg, w);
        break;
    case 10:
        hs_rank_store(lds, rg);
        break;
    case 11:
        hs_rank_out(t, lds, w);
        break;
    }
}
#define HS_PHASES 12

#endif
