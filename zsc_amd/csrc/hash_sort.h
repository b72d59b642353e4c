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
 * One workgroup of HS_WAVES wavefronts per tile of 32 768 positions.  Stable
 * LSD radix sort in two passes over the 15-bit hash (8 low bits, 7 high bits):
 *   phase 0  clear counters
 *   phase 1  per-wave-slice histogram of the low digit
 *   phase 2  exclusive scan (digit-major, wave-minor)
 *   phase 3  stable scatter into tmp: rank inside a 64-element step comes from
 *            ballot "multi-split" (one ballot per digit bit), not from atomics
 *   phase 4  per-wave-slice histogram of the high digit over tmp
 *   phase 5  scan
 *   phase 6  stable scatter into sorted[] + rank[]
 *   phase 7  bucket directory dir[h] = first sorted index with hash >= h
 * Counters are tiny (24 KiB of LDS), two tiles are resident per CU.
 *
 * Round 2 also built the sort wholly inside LDS (tile bytes, a 16-bit order array permuted in
 * place through registers, sorted[] / rank[] / dir[] written once and coalesced; commit
 * "hash sort staged wholly in LDS"): 41 ms per 1024 Canterbury-like sets with the ballot
 * multi-split, 49 ms with an LDS match table in its place, against 40 ms for this version --
 * the scattered accesses move from the texture path to LDS bank conflicts and cost the same
 * ~10 CU cycles per position.  At 145 G key-passes/s the sort already runs at the rate of
 * published one-sweep radix sorts; it stays as it was.
 *
 * Bytes per input byte (TILE positions): read 2 (input, twice) + tmp 4w+4r +
 * sorted 4w(+4r for dir) + rank 2w + dir 2w.
 */
#ifndef ZSC_HASH_SORT_H
#define ZSC_HASH_SORT_H

#include "wave.h"
#include "zsc_dev.h"

#define HS_WAVES 16

typedef struct {
    uint32_t cnt0[256 * HS_WAVES];
    uint32_t cnt1[128 * HS_WAVES];
} HsLds;

typedef struct {
    const uint8_t *in;  /* the tile's buffer */
    uint32_t n;         /* buffer length */
    uint32_t start;     /* absolute position of the tile's first byte */
    uint32_t m;         /* positions of this tile that own a 3-byte string (pos <= n-3) */
    uint32_t *sorted;   /* TILE entries */
    uint32_t *tmp;      /* TILE entries */
    uint16_t *rank;     /* per position of the buffer */
    uint16_t *dir;      /* DIR_STRIDE entries: first sorted index of every bucket */
    const uint16_t *dir_prev; /* directory of the previous tile of the same buffer, or null */
    uint16_t *hib;      /* per position of the buffer: last sorted index of its bucket in the previous tile */
    uint32_t *cnt;      /* per position of the buffer: earlier members of its bucket in its own tile (low
                           half) and members of the bucket in the previous tile (high half) */
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

DEV uint32_t hs_slice(uint32_t m)
{
    uint32_t per = (m + HS_WAVES * WAVE - 1) / (HS_WAVES * WAVE);
    return per * WAVE;
}

/* stable scatter of one pass; key_of_tmp selects pass 2 (source = tmp) */
DEV void hs_scatter(const HsTile &t, HsLds *lds, int w, int pass)
{
    const uint32_t slice = hs_slice(t.m);
    const uint32_t lo = (uint32_t)w * slice;
    const int nbits = pass == 0 ? 8 : 7;
    uint32_t *off = pass == 0 ? lds->cnt0 : lds->cnt1;

    for (uint32_t s = lo; s < lo + slice && s < t.m; s += WAVE) {
        LANEVAR(uint32_t, ent);
        LANEVAR(uint32_t, dig);
        LANEVAR(int, ok);
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            LV(ok) = i < t.m;
            uint32_t e = 0;
            if (LV(ok)) {
                if (pass == 0) {
                    uint32_t h = hs_hash3(t.in, t.start + i, t.n);
                    e = i | (h << 16);
                } else {
                    e = t.tmp[i];
                }
            }
            LV(ent) = e;
            LV(dig) = pass == 0 ? ((e >> 16) & 0xff) : (e >> 24);
        }
        /* ballot multi-split: lanes with the same digit find each other */
        LANEVAR(uint64_t, peers);
        uint64_t live = BALLOT(ok);
        FOR_LANES { LV(peers) = live; }
        for (int b = 0; b < nbits; b++) {
            LANEVAR(int, bit);
            FOR_LANES { LV(bit) = (int)((LV(dig) >> b) & 1u); }
            uint64_t ones = BALLOT(bit);
            FOR_LANES { LV(peers) &= LV(bit) ? ones : ~ones; }
        }
        LANEVAR(uint32_t, dst);
        FOR_LANES
        {
            if (LV(ok)) {
                uint32_t before = (uint32_t)POPC64(LV(peers) & ((1ull << LANE) - 1ull));
                LV(dst) = off[LV(dig) * HS_WAVES + (uint32_t)w] + before;
            }
        }
        FOR_LANES
        {
            if (LV(ok)) {
                uint64_t mine = LV(peers);
                if ((mine & ((1ull << LANE) - 1ull)) == 0) /* first lane of its digit group */
                    off[LV(dig) * HS_WAVES + (uint32_t)w] += (uint32_t)POPC64(mine);
            }
        }
        FOR_LANES
        {
            if (LV(ok)) {
                if (pass == 0) {
                    t.tmp[LV(dst)] = LV(ent);
                    /* the second pass counts its digits per wave slice of tmp: where this entry
                     * lands decides the slice, so that count is taken here and the counting
                     * pass over tmp is saved */
                    LDS_ADD_U32(&lds->cnt1[(LV(ent) >> 24) * HS_WAVES + LV(dst) / slice], 1u);
                } else {
                    t.sorted[LV(dst)] = LV(ent);
                    t.rank[t.start + (LV(ent) & ZD_TILE_MASK)] = (uint16_t)LV(dst);
                }
            }
        }
    }
}

DEV void hs_count(const HsTile &t, HsLds *lds, int w, int pass)
{
    const uint32_t slice = hs_slice(t.m);
    const uint32_t lo = (uint32_t)w * slice;
    uint32_t *cnt = pass == 0 ? lds->cnt0 : lds->cnt1;
    for (uint32_t s = lo; s < lo + slice && s < t.m; s += WAVE) {
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            if (i < t.m) {
                uint32_t d;
                if (pass == 0)
                    d = hs_hash3(t.in, t.start + i, t.n) & 0xff;
                else
                    d = t.tmp[i] >> 24;
                LDS_ADD_U32(&cnt[d * HS_WAVES + (uint32_t)w], 1u);
            }
        }
    }
}

/* exclusive scan of ndig*HS_WAVES counters, digit-major; done by wave 0 */
DEV void hs_scan(uint32_t *cnt, int ndig)
{
    uint32_t run = 0;
    const int total = ndig * HS_WAVES;
    for (int s = 0; s < total; s += WAVE) {
        LANEVAR(uint32_t, v);
        LANEVAR(uint32_t, ex);
        FOR_LANES { LV(v) = cnt[s + LANE]; }
        uint32_t tot;
        WAVE_EXSCAN(v, ex, tot);
        FOR_LANES { cnt[s + LANE] = run + LV(ex); }
        run += tot;
    }
}

DEV void hs_directory(const HsTile &t, int w)
{
    /* bucket starts: sorted index i opens every bucket in (h[i-1], h[i]] */
    for (uint32_t s = (uint32_t)w * WAVE; s < t.m; s += HS_WAVES * WAVE) {
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            if (i < t.m) {
                uint32_t h = t.sorted[i] >> 16;
                int32_t hp = i == 0 ? -1 : (int32_t)(t.sorted[i - 1] >> 16);
                for (int32_t hh = hp + 1; hh <= (int32_t)h; hh++)
                    t.dir[hh] = (uint16_t)i;
            }
        }
    }
    /* buckets past the last occupied one (and the end sentinel) point at m */
    int32_t hl = t.m == 0 ? -1 : (int32_t)(t.sorted[t.m - 1] >> 16);
    for (int32_t s = hl + 1 + w * WAVE; s <= 32768; s += HS_WAVES * WAVE) {
        FOR_LANES
        {
            int32_t hh = s + LANE;
            if (hh <= 32768)
                t.dir[hh] = (uint16_t)t.m;
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
                }
            }
        }
    }
}

/* one phase of the tile sort, executed by wave `w` of the tile's workgroup;
 * the caller puts a workgroup barrier between phases */
DEV void hash_sort_phase(const HsTile &t, HsLds *lds, int w, int phase)
{
    switch (phase) {
    case 0:
        for (int i = w * WAVE; i < 256 * HS_WAVES; i += HS_WAVES * WAVE) {
            FOR_LANES { lds->cnt0[i + LANE] = 0; }
        }
        for (int i = w * WAVE; i < 128 * HS_WAVES; i += HS_WAVES * WAVE) {
            FOR_LANES { lds->cnt1[i + LANE] = 0; }
        }
        break;
    case 1:
        hs_count(t, lds, w, 0);
        break;
    case 2:
        if (w == 0)
            hs_scan(lds->cnt0, 256);
        break;
    case 3:
        hs_scatter(t, lds, w, 0);
        break;
    case 4:
        break; /* counted while scattering (hs_scatter, pass 0) */
    case 5:
        if (w == 0)
            hs_scan(lds->cnt1, 128);
        break;
    case 6:
        hs_scatter(t, lds, w, 1);
        break;
    case 7:
        hs_directory(t, w);
        break;
    }
}
#define HS_PHASES 8

#endif
