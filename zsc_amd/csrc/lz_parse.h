/*
 * lz_parse.h -- kernel 2: the LZ77 parse, one wavefront per buffer.
 *
 * Restates deflate_slow (reference src/deflate.c:1989-2122) and longest_match
 * (:1400-1518) for a 64-lane wavefront.  The parse itself is inherently serial
 * (every decision depends on the previous match), so the wave keeps the parse
 * state in wave-uniform scalars and spends its 64 lanes on the part that costs
 * the reference 95 % of its time: walking hash chains.
 *
 *   - the chain of position p is a contiguous descending run of the tile's
 *     sorted array (hash_sort.h): one coalesced load brings 64 candidates;
 *   - every lane applies the reference's 4-byte pre-check (:1462-1465) to its
 *     own candidate against the 36 KiB LDS ring that holds the sliding window;
 *   - a ballot yields the candidates that pass, in chain order; each of those
 *     gets the full comparison (:1485-1488), done cooperatively (lane k
 *     compares bytes 4k..4k+3, a second ballot finds the first difference);
 *   - zsc's rule that pre-check failures are free and only full comparisons
 *     consume chain budget (:1467-1468 vs :1508-1509) falls out naturally: the
 *     failures of a batch are discarded by the ballot, 64 at a time.
 *
 * Positions are absolute; the reference's window slides are tracked as the
 * number `base` (the formulation the CPU checker under oracle/ pins against the
 * reference).  No hash table is updated while parsing, so the positions covered
 * by an emitted match cost nothing.
 */
#ifndef ZSC_LZ_PARSE_H
#define ZSC_LZ_PARSE_H

#include "wave.h"
#include "zsc_dev.h"

typedef struct {
    uint8_t ring[ZD_RING + 512]; /* sliding window; first 16 bytes mirrored after the end, rest slack for masked over-reads */
    uint32_t stage[WAVE];       /* symbols waiting for a coalesced store */
} LzLds;

typedef struct {
    const uint8_t *in;      /* this buffer */
    uint32_t n;
    const uint32_t *sorted; /* tile 0 of this buffer; tile t at + t*ZD_TILE */
    const uint16_t *rank;   /* this buffer */
    const uint16_t *dir;    /* tile 0 of this buffer; tile t at + t*ZD_DIR_STRIDE */
    uint32_t *syms;         /* this buffer's symbol slots */
    ZdBlockRec *blocks;     /* this buffer's block records */
    ZdParseOut *out;
    ZdLevel cfg;
    uint32_t strategy;
} LzJob;

/* wave-uniform parser state */
typedef struct {
    uint32_t lo, hi;    /* absolute positions held by the ring: [lo, hi) */
    uint32_t wrap_base; /* multiple of ZD_RING with lo - wrap_base < ZD_RING */
    uint32_t base;      /* reference window base (multiple of 32768) */
    uint32_t data_end;  /* end of the data the reference's window would hold */
    uint32_t nsyms, nstaged;
    uint32_t nblocks, blk_sym0, blk_in0;
} LzState;

DEV uint32_t lz_ridx(const LzState &st, uint32_t pos)
{
    uint32_t r = pos - st.wrap_base;
    return r >= ZD_RING ? r - ZD_RING : r;
}

/* bring [hi, hi+CHUNK) into the ring (chunk-aligned, 16 bytes per lane per step) */
DEV void lz_load_chunk(const LzJob &job, LzLds *lds, LzState &st)
{
    const uint32_t a0 = st.hi; /* multiple of ZD_CHUNK */
    const uint32_t r0 = lz_ridx(st, a0);
    for (uint32_t k = 0; k < ZD_CHUNK; k += WAVE * 16) {
        FOR_LANES
        {
            uint32_t off = k + (uint32_t)LANE * 16u;
            uint32_t a = a0 + off;
            uint8_t *dst = &lds->ring[r0 + off];
            if (a + 16 <= job.n) {
                /* 16-byte aligned: buffers start 16-byte aligned in the batch */
                COPY16(dst, job.in + a);
            } else {
                for (uint32_t j = 0; j < 16; j++)
                    dst[j] = a + j < job.n ? job.in[a + j] : (uint8_t)0;
            }
        }
    }
    if (r0 == 0) {
        FOR_LANES
        {
            if (LANE < 16)
                lds->ring[ZD_RING + LANE] = lds->ring[LANE];
        }
    }
    st.hi = a0 + ZD_CHUNK;
    if (st.hi - st.lo > ZD_RING)
        st.lo = st.hi - ZD_RING;
    if (st.lo - st.wrap_base >= ZD_RING)
        st.wrap_base += ZD_RING;
}

/* make sure the ring holds everything position p can touch: p-32506 .. p+261 */
DEV void lz_ensure(const LzJob &job, LzLds *lds, LzState &st, uint32_t p)
{
    while (st.hi < job.n && st.hi < p + ZD_MIN_LOOKAHEAD)
        lz_load_chunk(job, lds, st);
}

DEV void lz_flush_stage(const LzJob &job, LzLds *lds, LzState &st)
{
    const uint32_t first = st.nsyms - st.nstaged;
    FOR_LANES
    {
        if ((uint32_t)LANE < st.nstaged)
            job.syms[first + (uint32_t)LANE] = lds->stage[LANE];
    }
    st.nstaged = 0;
}

/* _tr_tally_*, reference include/zsc/deflate.h:338-354; returns "block is full" */
DEV int lz_put(const LzJob &job, LzLds *lds, LzState &st, uint32_t sym)
{
    ON_LANE0 { lds->stage[st.nstaged] = sym; }
    st.nstaged++;
    st.nsyms++;
    if (st.nstaged == WAVE)
        lz_flush_stage(job, lds, st);
    return st.nsyms - st.blk_sym0 == ZD_SYM_CAP;
}

/* FLUSH_BLOCK_ONLY, reference src/deflate.c:1660-1668 */
DEV void lz_cut(const LzJob &job, LzState &st, uint32_t upto, uint32_t last)
{
    ON_LANE0
    {
        ZdBlockRec *b = &job.blocks[st.nblocks];
        b->sym_begin = st.blk_sym0;
        b->sym_count = st.nsyms - st.blk_sym0;
        b->in_begin = st.blk_in0;
        b->in_len = upto - st.blk_in0;
        b->stored_ok = st.blk_in0 >= st.base ? 1u : 0u;
        b->last = last;
    }
    st.nblocks++;
    st.blk_sym0 = st.nsyms;
    st.blk_in0 = upto;
}

/* fill_window's slide decision, reference src/deflate.c:1563-1570,1589 */
DEV void lz_refill(const LzJob &job, LzState &st, uint32_t p)
{
    if (p - st.base >= ZD_TILE + ZD_MAX_DIST)
        st.base += ZD_TILE;
    uint64_t end = (uint64_t)st.base + 2ull * ZD_TILE;
    st.data_end = end < job.n ? (uint32_t)end : job.n;
}

/* cooperative longest-common-prefix of the strings at q and p, at most cap (<=258) bytes */
DEV uint32_t lz_lcp(const LzLds *lds, const LzState &st, uint32_t q, uint32_t p, uint32_t cap)
{
    LANEVAR(uint32_t, diff);
    LANEVAR(int, differs);
    FOR_LANES
    {
        uint32_t a = ld_u32(&lds->ring[lz_ridx(st, q + 4u * (uint32_t)LANE)]);
        uint32_t b = ld_u32(&lds->ring[lz_ridx(st, p + 4u * (uint32_t)LANE)]);
        LV(diff) = a ^ b;
        LV(differs) = LV(diff) != 0;
    }
    uint64_t m = BALLOT(differs);
    uint32_t len;
    if (m != 0) {
        int f = CTZ64(m);
        uint32_t x = READLANE(diff, f);
        len = 4u * (uint32_t)f + ((uint32_t)CTZ32(x) >> 3);
    } else {
        len = 256;
        if (cap > 256 && lds->ring[lz_ridx(st, q + 256)] == lds->ring[lz_ridx(st, p + 256)]) {
            len = 257;
            if (cap > 257 && lds->ring[lz_ridx(st, q + 257)] == lds->ring[lz_ridx(st, p + 257)])
                len = 258;
        }
    }
    return len < cap ? len : cap;
}

/* longest_match over the sorted runs.  Returns 0 when the reference would not have
 * called longest_match at all (no live chain head, src/deflate.c:2027-2028),
 * otherwise the value longest_match returns; *where as in s->match_start. */
DEV uint32_t lz_search(const LzJob &job, const LzLds *lds, const LzState &st, uint32_t p,
                       uint32_t prev_len, uint32_t *where)
{
    const uint32_t look = st.data_end - p;
    const uint32_t tile = p >> 15;
    const uint32_t rp = lz_ridx(st, p);
    const uint32_t w0 = ld_u32(&lds->ring[rp]);
    const uint32_t h = (((w0 & 0xff) << 10) ^ (((w0 >> 8) & 0xff) << 5) ^ ((w0 >> 16) & 0xff)) & ZD_HASH_MASK;
    const uint32_t s01 = w0 & 0xffff;

    uint32_t best = prev_len;
    uint32_t budget = job.cfg.chain;
    uint32_t nice = job.cfg.nice;
    const uint32_t cap = look < 258u ? look : 258u;
    if (prev_len >= job.cfg.good)
        budget >>= 2;
    if (nice > look)
        nice = look;

    int head_seen = 0;
    uint32_t sb = 0; /* bytes best-1, best of the scan string */

    for (int seg = 0; seg < 2; seg++) {
        const uint32_t *run;
        uint32_t tile_pos;
        int32_t hi_idx, lo_idx;
        if (seg == 0) {
            run = job.sorted + (uint64_t)tile * ZD_TILE;
            tile_pos = tile << 15;
            hi_idx = (int32_t)job.rank[p] - 1;
            lo_idx = 0;
        } else {
            if (tile == 0)
                break;
            const uint16_t *d = job.dir + (uint64_t)(tile - 1) * ZD_DIR_STRIDE;
            run = job.sorted + (uint64_t)(tile - 1) * ZD_TILE;
            tile_pos = (tile - 1) << 15;
            lo_idx = (int32_t)d[h];
            hi_idx = (int32_t)d[h + 1] - 1;
        }
        while (hi_idx >= lo_idx) {
            LANEVAR(uint32_t, q);
            LANEVAR(int, inb);   /* belongs to the chain (same hash bucket) */
            LANEVAR(int, alive); /* and is still reachable from p */
            FOR_LANES
            {
                int32_t i = hi_idx - LANE;
                uint32_t e = i >= lo_idx ? run[i] : ZD_ENTRY_NONE;
                LV(inb) = (e >> 16) == h;
                LV(q) = tile_pos + (e & ZD_TILE_MASK);
            }
            uint64_t m_in = BALLOT(inb);
            if (!head_seen && !(m_in & 1ull))
                break; /* nothing of this hash in this tile */
            FOR_LANES
            {
                uint32_t d = p - LV(q);
                int near = d < ZD_MAX_DIST || (d == ZD_MAX_DIST && !head_seen && LANE == 0);
                LV(alive) = LV(inb) && LV(q) > st.base && near;
            }
            uint64_t m_alive = BALLOT(alive);
            if (!head_seen) {
                if (!(m_alive & 1ull))
                    return 0; /* chain head is NIL or too far: no call */
                head_seen = 1;
                if (best >= look)
                    return look;
                sb = ld_u16(&lds->ring[lz_ridx(st, p + best - 1)]);
            }
            const int ends_here = (m_in & ~m_alive) != 0; /* a chain member out of reach */

            LANEVAR(int, pass);
            FOR_LANES
            {
                int c = 0;
                if (LV(alive)) {
                    c = ld_u16(&lds->ring[lz_ridx(st, LV(q))]) == s01 &&
                        ld_u16(&lds->ring[lz_ridx(st, LV(q) + best - 1)]) == sb;
                }
                LV(pass) = c;
            }
            uint64_t todo = BALLOT(pass);
            while (todo != 0) {
                const int j = CTZ64(todo);
                const uint32_t qj = READLANE(q, j);
                const uint32_t len = lz_lcp(lds, st, qj, p, cap);
                if (len > best) {
                    *where = qj;
                    best = len;
                    if (len >= nice)
                        return best < look ? best : look;
                    if (--budget == 0)
                        return best < look ? best : look;
                    sb = ld_u16(&lds->ring[lz_ridx(st, p + best - 1)]);
                    FOR_LANES
                    {
                        int c = 0;
                        if (LV(alive) && LANE > j) {
                            c = ld_u16(&lds->ring[lz_ridx(st, LV(q))]) == s01 &&
                                ld_u16(&lds->ring[lz_ridx(st, LV(q) + best - 1)]) == sb;
                        }
                        LV(pass) = c;
                    }
                    todo = BALLOT(pass);
                } else {
                    if (--budget == 0)
                        return best < look ? best : look;
                    todo &= todo - 1;
                }
            }
            if (ends_here)
                return best < look ? best : look;
            if (m_in != ~0ull)
                break; /* bucket (or tile) exhausted: continue in the older tile */
            hi_idx -= WAVE;
        }
    }
    if (!head_seen)
        return 0;
    return best < look ? best : look;
}

/* deflate_slow, reference src/deflate.c:1989-2122, flush == Z_FINISH */
DEV void lz_parse_lazy(const LzJob &job, LzLds *lds)
{
    LzState st;
    st.lo = st.hi = st.wrap_base = 0;
    st.base = 0;
    st.data_end = 0;
    st.nsyms = st.nstaged = 0;
    st.nblocks = st.blk_sym0 = st.blk_in0 = 0;

    uint32_t p = 0, cur_len = 2, cur_at = 0;
    int pending = 0; /* match_available */

    for (;;) {
        uint32_t look = st.data_end - p;
        if (look < ZD_MIN_LOOKAHEAD) {
            lz_refill(job, st, p);
            look = st.data_end - p;
            if (look == 0)
                break;
        }
        lz_ensure(job, lds, st, p);

        const uint32_t prev_len = cur_len, prev_at = cur_at;
        cur_len = 2;
        if (look >= 3 && prev_len < job.cfg.lazy) {
            uint32_t got = lz_search(job, lds, st, p, prev_len, &cur_at);
            if (got != 0) {
                cur_len = got;
                if (cur_len <= 5 &&
                    (job.strategy == 1 || (cur_len == 3 && p - cur_at > ZD_TOO_FAR)))
                    cur_len = 2;
            }
        }
        if (prev_len >= 3 && cur_len <= prev_len) {
            const int full = lz_put(job, lds, st, ((p - 1 - prev_at) << 16) | (prev_len - 3));
            pending = 0;
            cur_len = 2;
            p += prev_len - 1;
            if (full)
                lz_cut(job, st, p, 0);
        } else if (pending) {
            const uint32_t c = lds->ring[lz_ridx(st, p - 1)];
            if (lz_put(job, lds, st, c))
                lz_cut(job, st, p, 0);
            p++;
        } else {
            pending = 1;
            p++;
        }
    }
    if (pending)
        (void)lz_put(job, lds, st, lds->ring[lz_ridx(st, p - 1)]);
    lz_cut(job, st, p, 1);
    if (st.nstaged)
        lz_flush_stage(job, lds, st);
    ON_LANE0
    {
        job.out->nsyms = st.nsyms;
        job.out->nblocks = st.nblocks;
    }
}

#endif
