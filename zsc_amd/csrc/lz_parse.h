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

#define LZ_PR 512u /* positions covered by the rank/hib look-ahead ring */
#define LZ_MIRROR 512u /* bytes of the window ring's start kept a second time behind its end */
#define LZ_PR_FAST 128u /* the same in the greedy parser's LDS */

/* LDS of the lazy parser (levels 4-9): 36 KiB window ring + look-ahead rings = 39 680 B,
 * four waves per CU */
template <uint32_t RING_BYTES, uint32_t CHUNK_BYTES, bool WITH_HOLES = false>
struct LzLdsT {
    static constexpr uint32_t RING = RING_BYTES, CHUNK = CHUNK_BYTES;
    static constexpr bool HAS_INS = WITH_HOLES; /* lz_load_chunk clears ins[] for what enters the ring */
    static constexpr bool HOLES = WITH_HOLES;
    static constexpr bool GLOBAL_WIN = false; /* the window is the LDS ring */
    static constexpr uint32_t PR = LZ_PR;
    uint8_t ring[RING + 512]; /* sliding window; the first LZ_MIRROR bytes mirrored after the end, rest slack for masked over-reads */
    uint32_t stage[WAVE];     /* symbols waiting for a coalesced store */
    uint16_t prank[LZ_PR];    /* rank[] of the next few hundred positions */
    uint16_t phib[LZ_PR];     /* hib[] of the same positions */
    /* WITH_HOLES (runs of sections with joints): bit r = the position at ring index r never
     * entered the reference's hash chains (lz_parse_lazy, "holes") */
    uint32_t ins[WITH_HOLES ? RING / 32 : 1];
};
typedef LzLdsT<ZD_RING, ZD_CHUNK> LzLds;
/* a buffer that fits the ring whole never slides it, so short buffers can run with a
 * smaller ring and proportionally more waves per CU (the parser is latency-bound) */
typedef LzLdsT<18432u, 2048u> LzLds16k; /* n <= 18 432: 7 waves per CU */
typedef LzLdsT<10240u, 2048u> LzLds8k;  /* n <= 10 240: 12 waves per CU */
typedef LzLdsT<6144u, 2048u> LzLds4k;   /* n <=  6 144: 17 waves per CU */
/* the same with the hole map, for runs of sections that have joints */
typedef LzLdsT<ZD_RING, ZD_CHUNK, true> LzLdsJ;
typedef LzLdsT<18432u, 2048u, true> LzLdsJ16k;
typedef LzLdsT<10240u, 2048u, true> LzLdsJ8k;
typedef LzLdsT<6144u, 2048u, true> LzLdsJ4k;

/* LDS of the greedy parser (levels 1-3): a 34 KiB ring (2 KiB chunks) leaves room for
 * the one-bit-per-position "was inserted" map deflate_fast needs (it does not index the
 * inside of long matches, src/deflate.c:1940-1962) = 39 936 B, four waves per CU */
struct LzLdsFast {
    static constexpr uint32_t RING = 34816u, CHUNK = 2048u;
    static constexpr bool HAS_INS = true;
    static constexpr bool HOLES = false;
    static constexpr bool GLOBAL_WIN = false;
    static constexpr uint32_t PR = LZ_PR_FAST;
    uint8_t ring[RING + 512];
    uint32_t stage[WAVE];
    uint32_t ins[RING / 32];  /* bit r: the position at ring index r is in the hash chains */
    /* rank[] / hib[] of the next positions, 64 per coalesced load (as the lazy parser's prank / phib,
     * but 128 entries: with 512 the fifth KiB would cost the fourth wave of a CU) -- reading them from
     * global memory position by position was a dependent round trip per searched position */
    uint16_t prank[LZ_PR_FAST];
    uint16_t phib[LZ_PR_FAST];
};

/* The greedy parser without a window in LDS.  At levels 1-3 a search looks at a handful of
 * candidates (max_chain 4 .. 32, nice 8 .. 32): their bytes come through the caches from the input
 * itself, and what is left in LDS is the "was inserted" map (one bit per position of a 40 KiB span)
 * and the staging areas -- 6 KiB instead of 40, so a CU holds 24 parsers instead of 4 (the parse
 * is a chain of latencies per position: what it needs is more of them in flight). */
struct LzLdsFastG {
    static constexpr uint32_t RING = 40960u, CHUNK = 2048u; /* (the span of the map: the window and what lies ahead of it; nothing is copied) */
    static constexpr bool HAS_INS = true;
    static constexpr bool HOLES = false;
    static constexpr bool GLOBAL_WIN = true;
    static constexpr uint32_t PR = LZ_PR_FAST;
    uint8_t ring[16]; /* (not used) */
    uint32_t stage[WAVE];
    uint32_t ins[RING / 32];
    uint16_t prank[LZ_PR_FAST];
    uint16_t phib[LZ_PR_FAST];
};

typedef struct {
    const uint8_t *in;      /* this buffer */
    uint32_t n;
    const uint32_t *sorted; /* tile 0 of this buffer; tile t at + t*ZD_TILE */
    const uint16_t *rank;   /* this buffer: index of a position in its tile's sorted array */
    const uint16_t *hib;    /* this buffer: last index of the position's bucket in the previous tile */
    const uint32_t *cnt;    /* this buffer: chain lengths in the own | previous tile (hash_sort.h) */
    const uint16_t *dir;    /* segmented parser: the bucket directories of this buffer's tiles (ZD_DIR_STRIDE apart); then
                               hib / cnt are not read -- the parser works them out itself (sg_link), k_link_prev need not run */
    uint32_t stair_min;     /* segmented parser: chains at least this long are searched as a staircase (lz_parse_seg.h) */
    const uint32_t *r2;     /* this buffer: the match table, longest_match(p, 2) per position (match_table.h), or null */
    uint32_t *syms;         /* this buffer's symbol slots */
    ZdBlockRec *blocks;     /* this buffer's block records */
    ZdParseOut *out;
    ZdLevel cfg;
    uint32_t strategy;
    uint32_t more;        /* ZdBuf.more */
    const ZdSched *sched; /* joints of a run of sections (zsc_dev.h), in the order they are met */
    uint32_t nsched;
    uint32_t n0;          /* with joints: the length of the run's first section */
    uint32_t ntot;        /* bytes behind `in` (= n, except while the segmented parser works through
                             the phases of a run with joints: then n is the phase's end) */
} LzJob;

/* wave-uniform parser state */
typedef struct {
    uint32_t lo, hi;    /* absolute positions held by the ring: [lo, hi) */
    uint32_t wrap_base; /* multiple of ZD_RING with lo - wrap_base < ZD_RING */
    uint32_t base;      /* reference window base (multiple of 32768) */
    uint32_t data_end;  /* end of the data the reference's window would hold */
    uint32_t nsyms, nstaged;
    uint32_t nblocks, blk_sym0, blk_in0;
    uint32_t pr_hi;     /* rank/hib are staged in LDS for positions below this */
    uint32_t n;         /* input deflate() has been given so far (job.n unless the run has joints) */
    uint32_t si;        /* next joint */
    uint32_t it;        /* position of the current parse-loop iteration */
} LzState;

template <class L>
DEV uint32_t lz_ridx(const LzState &st, uint32_t pos)
{
    uint32_t r = pos - st.wrap_base;
    return r >= L::RING ? r - L::RING : r;
}

/* four / one byte(s) of the window at an absolute position: from the LDS ring, or -- a parser
 * without one -- from the input itself, bytes behind it reading as zero like the ring's */
template <class L>
DEV uint32_t lz_w32(const LzJob &job, const L *lds, const LzState &st, uint32_t pos)
{
    if constexpr (L::GLOBAL_WIN) {
        (void)lds;
        (void)st;
        if ((uint64_t)pos + 4u <= job.ntot)
            return ld_u32(job.in + pos);
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4u; k++)
            v |= (pos + k < job.ntot ? (uint32_t)job.in[pos + k] : 0u) << (8u * k);
        return v;
    } else {
        (void)job;
        return lds_u32(lds->ring, lz_ridx<L>(st, pos));
    }
}
template <class L>
DEV uint32_t lz_w8(const LzJob &job, const L *lds, const LzState &st, uint32_t pos)
{
    if constexpr (L::GLOBAL_WIN) {
        (void)lds;
        (void)st;
        return pos < job.ntot ? (uint32_t)job.in[pos] : 0u;
    } else {
        (void)job;
        return lds->ring[lz_ridx<L>(st, pos)];
    }
}

/* bring [hi, hi+CHUNK) into the ring (chunk-aligned, 16 bytes per lane per step) */
template <class L>
DEV void lz_load_chunk(const LzJob &job, L *lds, LzState &st)
{
    const uint32_t a0 = st.hi; /* multiple of CHUNK */
    const uint32_t r0 = lz_ridx<L>(st, a0);
    for (uint32_t k = 0; k < L::CHUNK && !L::GLOBAL_WIN; k += WAVE * 16) {
        FOR_LANES
        {
            uint32_t off = k + (uint32_t)LANE * 16u;
            uint32_t a = a0 + off;
            uint8_t *dst = &lds->ring[r0 + off];
            if (a + 16 <= job.ntot) {
                /* 16-byte aligned: buffers start 16-byte aligned in the batch */
                COPY16(dst, job.in + a);
            } else {
                for (uint32_t j = 0; j < 16; j++)
                    dst[j] = a + j < job.ntot ? job.in[a + j] : (uint8_t)0;
            }
        }
    }
    if constexpr (L::HAS_INS) {
        /* positions that enter the ring have not been inserted into any chain yet */
        for (uint32_t k = 0; k < L::CHUNK / 32u; k += WAVE) {
            FOR_LANES
            {
                if (k + (uint32_t)LANE < L::CHUNK / 32u)
                    lds->ins[r0 / 32u + k + (uint32_t)LANE] = 0;
            }
        }
    }
    WAVE_SYNC();
    if (r0 == 0 && !L::GLOBAL_WIN) {
        /* the first LZ_MIRROR bytes again behind the end: a string that starts inside the ring is
         * read with linear indices, whatever its length (at most MAX_MATCH and a few bytes) */
        for (uint32_t k = 0; k < LZ_MIRROR; k += 4u * WAVE) {
            FOR_LANES
            {
                const uint32_t o = k + 4u * (uint32_t)LANE;
                if (o < LZ_MIRROR)
                    *(uint32_t *)&lds->ring[L::RING + o] = *(const uint32_t *)&lds->ring[o];
            }
        }
    }
    WAVE_SYNC();
    st.hi = a0 + L::CHUNK;
    if (st.hi - st.lo > L::RING)
        st.lo = st.hi - L::RING;
    if (st.lo - st.wrap_base >= L::RING)
        st.wrap_base += L::RING;
}

/* make sure the ring holds everything position p can touch: p-32506 .. p+261 */
template <class L>
DEV void lz_ensure(const LzJob &job, L *lds, LzState &st, uint32_t p)
{
    while (st.hi < job.ntot && st.hi < p + ZD_MIN_LOOKAHEAD)
        lz_load_chunk<L>(job, lds, st);
}

template <class L>
DEV void lz_flush_stage(const LzJob &job, L *lds, LzState &st)
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
template <class L>
DEV int lz_put(const LzJob &job, L *lds, LzState &st, uint32_t sym)
{
    ON_LANE0 { lds->stage[st.nstaged] = sym; }
    WAVE_SYNC();
    st.nstaged++;
    st.nsyms++;
    if (st.nstaged == WAVE)
        lz_flush_stage<L>(job, lds, st);
    return st.nsyms - st.blk_sym0 == job.cfg.sym_cap;
}

/* FLUSH_BLOCK_ONLY, reference src/deflate.c:1660-1668 */
/* joints of kind 0 that can be folded into the phase that starts now (zsc_dev.h) */
DEV void lz_fold(const LzJob &job, LzState &st)
{
    while (st.si < job.nsched && UNI(job.sched[st.si].kind) == 0u) {
        const uint32_t jp = UNI(job.sched[st.si].pos);
        if (jp != ZD_JOINT_ANYWHERE && (uint64_t)jp + ZD_MIN_LOOKAHEAD > st.n)
            break; /* too close to the end: lz_cut switches at the cut itself */
        st.n = UNI(job.sched[st.si].new_n);
        st.si++;
    }
}

/* returns the end of the input as it was before, if a joint of kind 0 let the next section in
 * at this cut; 0xffffffff otherwise */
DEV uint32_t lz_cut(const LzJob &job, LzState &st, uint32_t upto, uint32_t last, uint32_t cut)
{
    uint32_t joint_n = 0xffffffffu;
    ON_LANE0
    {
        ZdBlockRec *b = &job.blocks[st.nblocks];
        b->sym_begin = st.blk_sym0;
        b->sym_count = st.nsyms - st.blk_sym0;
        b->in_begin = st.blk_in0;
        b->in_len = upto - st.blk_in0;
        b->stored_ok = st.blk_in0 >= st.base ? 1u : 0u;
        b->last = last;
        b->cut = cut;
        b->at = cut == ZD_CUT_END ? upto : st.it;
        const uint64_t wend = (uint64_t)st.base + 2ull * job.cfg.wsize;
        b->wend = wend < 0xffffffffull ? (uint32_t)wend : 0xffffffffu;
    }
    st.nblocks++;
    st.blk_sym0 = st.nsyms;
    st.blk_in0 = upto;
    /* a joint of kind 0: the next section was let in while this block was being flushed */
    if (cut == ZD_CUT_FULL && st.si < job.nsched) {
        const uint32_t jp = UNI(job.sched[st.si].pos), jk = UNI(job.sched[st.si].kind);
        if (jk == 0u && jp == upto) {
            joint_n = st.n;
            st.n = UNI(job.sched[st.si].new_n);
            st.si++;
            lz_fold(job, st);
        }
    }
    return joint_n;
}

/* the input given so far is used up at p: is there a joint of kind 1 here?  (The caller has
 * sent the owed literal.)  Cuts the block if it holds anything (src/deflate.c:2118-2120) and
 * lets the next section in. */
DEV int lz_joint_at_end(const LzJob &job, LzState &st, uint32_t p)
{
    if (st.si >= job.nsched)
        return 0;
    if (UNI(job.sched[st.si].kind) != 1u || UNI(job.sched[st.si].pos) != p)
        return 0;
    if (st.nsyms != st.blk_sym0)
        (void)lz_cut(job, st, p, 0, ZD_CUT_END);
    st.n = UNI(job.sched[st.si].new_n);
    st.si++;
    lz_fold(job, st);
    return 1;
}

/* the end of the run: FLUSH_BLOCK(s, 1) when the stream ends here (:2114-2117), else the
 * block only if it holds anything (:2118-2120) */
DEV void lz_cut_end(const LzJob &job, LzState &st, uint32_t p)
{
    if (!job.more)
        (void)lz_cut(job, st, p, 1, ZD_CUT_END);
    else if (st.nsyms != st.blk_sym0)
        (void)lz_cut(job, st, p, 0, ZD_CUT_END);
}

/* fill_window's slide decision, reference src/deflate.c:1563-1570,1589 */
DEV void lz_refill(const LzJob &job, LzState &st, uint32_t p)
{
    if (p - st.base >= job.cfg.wsize + job.cfg.max_dist)
        st.base += job.cfg.wsize;
    uint64_t end = (uint64_t)st.base + 2ull * job.cfg.wsize;
    st.data_end = end < st.n ? (uint32_t)end : st.n;
}

/* cooperative longest-common-prefix of the strings at q and p, at most cap (<=258) bytes */
template <class L>
DEV uint32_t lz_lcp(const LzJob &job, const L *lds, const LzState &st, uint32_t q, uint32_t p, uint32_t cap)
{
    LANEVAR(uint32_t, diff);
    LANEVAR(int, differs);
    FOR_LANES
    {
        uint32_t a = lz_w32<L>(job, lds, st, (q + 4u * (uint32_t)LANE));
        uint32_t b = lz_w32<L>(job, lds, st, (p + 4u * (uint32_t)LANE));
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
        if (cap > 256 && UNI(lz_w8<L>(job, lds, st, q + 256)) == UNI(lz_w8<L>(job, lds, st, p + 256))) {
            len = 257;
            if (cap > 257 && UNI(lz_w8<L>(job, lds, st, q + 257)) == UNI(lz_w8<L>(job, lds, st, p + 257)))
                len = 258;
        }
    }
    return len < cap ? len : cap;
}

/* stage rank[]/hib[] of the positions ahead of p in LDS (64 per step, coalesced) */
template <class L>
DEV void lz_ensure_ranks(const LzJob &job, L *lds, LzState &st, uint32_t p)
{
    while (st.pr_hi < job.ntot && st.pr_hi < p + 384u) {
        FOR_LANES
        {
            uint32_t x = st.pr_hi + (uint32_t)LANE;
            lds->prank[x & (LZ_PR - 1)] = job.rank[x];
            lds->phib[x & (LZ_PR - 1)] = job.hib[x];
        }
        WAVE_SYNC();
        st.pr_hi += WAVE;
    }
}

/* Issue the loads for the first 64 candidates of position x in its own tile (EA)
 * and in the previous tile (EB).  Called one search ahead, so the latency of these
 * loads overlaps the search that runs meanwhile. */
#define LZ_FETCH_FIRST(x, EA, EB)                                                             \
    do {                                                                                      \
        const uint32_t _t = (x) >> 15;                                                        \
        const uint32_t *_ra = job.sorted + (uint64_t)_t * ZD_TILE;                            \
        const int32_t _ha = (int32_t)UNI(lds->prank[(x) & (L::PR - 1)]) - 1;                  \
        const int32_t _hb = _t ? (int32_t)(int16_t)UNI(lds->phib[(x) & (L::PR - 1)]) : -1;    \
        FOR_LANES                                                                             \
        {                                                                                     \
            int32_t _ia = _ha - LANE, _ib = _hb - LANE;                                       \
            LV(EA) = _ia >= 0 ? _ra[_ia] : ZD_ENTRY_NONE;                                     \
            LV(EB) = _ib >= 0 ? (_ra - ZD_TILE)[_ib] : ZD_ENTRY_NONE;                         \
        }                                                                                     \
    } while (0)

/* The one place where the reference's hash function shows through.  A candidate at exactly
 * MAX_DIST is only looked at when it is the HEAD of p's chain (src/deflate.c:2027-2028 tests
 * the head with <=, the walk :1512 later links with <).  Chains here are those of the 15-bit
 * hash of mem_level 8; with another mem_level the reference's chains collide differently,
 * and the candidate is the head only if no position between it and p has p's hash under
 * THAT function (hash_bits = mem_level + 7, hash_shift = (hash_bits + 2) / 3, :343-350).
 * Everywhere else collisions cannot matter: a candidate that does not start with p's three
 * bytes fails the pre-check, and failing is free in zsc.  `MEMB` as in LZ_EVAL_BATCH. */
#define LZ_HEAD_BLOCKED(Q, P, MEMB, blocked)                                                  \
    do {                                                                                      \
        (blocked) = 0;                                                                        \
        const uint32_t _hs = (job.cfg.hbits + 2u) / 3u, _hm = (1u << job.cfg.hbits) - 1u;     \
        const uint32_t _wp = UNI(lz_w32<L>(job, lds, st, ((P))));                    \
        const uint32_t _hp = (((_wp & 0xffu) << (2u * _hs)) ^ (((_wp >> 8) & 0xffu) << _hs) ^ \
                              ((_wp >> 16) & 0xffu)) & _hm;                                   \
        for (uint32_t _x0 = (Q) + 1u; _x0 < (P) && !(blocked); _x0 += WAVE) {                 \
            LANEVAR(int, _same);                                                              \
            FOR_LANES                                                                         \
            {                                                                                 \
                const uint32_t _x = _x0 + (uint32_t)LANE;                                     \
                int _s = 0;                                                                   \
                if (_x < (P)) {                                                               \
                    const uint32_t _w = lz_w32<L>(job, lds, st, (_x));               \
                    const uint32_t _h = (((_w & 0xffu) << (2u * _hs)) ^                       \
                                         (((_w >> 8) & 0xffu) << _hs) ^ ((_w >> 16) & 0xffu)) & _hm; \
                    _s = _h == _hp && MEMB(_x);                                               \
                }                                                                             \
                LV(_same) = _s;                                                               \
            }                                                                                 \
            if (BALLOT(_same) != 0)                                                           \
                (blocked) = 1;                                                                \
        }                                                                                     \
    } while (0)

/* four bytes of the string being searched for, as a wave-uniform value: from the window, unless the
 * parser keeps them in registers (lz_parse_greedy redefines this) */
#define LZ_SPEEK32(pos) UNI(lz_w32<L>(job, lds, st, (pos)))

/* search context shared by the batch evaluator */
typedef struct {
    uint32_t p, h, s01, sb, look, cap, nice;
    uint32_t best, budget, where;
    int head_seen;
} LzSearch;

/* Holes (runs of sections with joints, lazy parse).  deflate_slow indexes every position -- but
 * only once three bytes of lookahead are there (:2018, and max_insert inside a match, :2069-2075).
 * The last two positions of the input given so far are caught up with by fill_window when the
 * call ended regularly (s->insert, :2113 + :1591-1612).  When deflate() instead came back from
 * flushing a full block right there and the next section was let in (a joint of kind 0 at the
 * very end of the input), nobody catches up: what the parse had already passed of those two
 * positions never enters the chains.  One bit per ring position remembers them. */
template <class L>
DEV int lz_is_hole(const L *lds, const LzState &st, uint32_t q)
{
    if constexpr (L::HOLES) {
        const uint32_t r = lz_ridx<L>(st, q);
        return (int)((lds->ins[r >> 5] >> (r & 31u)) & 1u);
    } else {
        (void)lds;
        (void)st;
        (void)q;
        return 0;
    }
}

template <class L>
DEV void lz_mark_holes(L *lds, const LzState &st, uint32_t old_n, uint32_t p_next)
{
    if constexpr (L::HOLES) {
        for (uint32_t k = 2; k >= 1; k--) {
            if (old_n >= k && old_n - k < p_next) {
                const uint32_t r = lz_ridx<L>(st, old_n - k);
                ON_LANE0 { lds->ins[r >> 5] |= 1u << (r & 31u); }
            }
        }
        WAVE_SYNC();
    } else {
        (void)lds;
        (void)st;
        (void)old_n;
        (void)p_next;
    }
}

/* chain membership of a candidate: the lazy parser indexes every position
 * (src/deflate.c:2018,2069-2075), the greedy one only those it marked (:1914,1940-1950) */
#define LZ_MEMB_ALL(q) 1
#define LZ_MEMB_LAZY(q) (!lz_is_hole<L>(lds, st, (q)))
#define LZ_MEMB_INS(q) ((lds->ins[lz_ridx<L>(st, (q)) >> 5] >> (lz_ridx<L>(st, (q)) & 31u)) & 1u)

/* Evaluate one batch of 64 sorted entries (ENT, newest first) that lie in the tile
 * starting at TPOS.  Sets `verdict`: 0 continue with the next batch of this run,
 * 1 this run is exhausted, 2 search finished (result in sc), 3 no live chain head. */
#define LZ_EVAL_BATCH(ENT, TPOS, MEMB, verdict)                                               \
    do {                                                                                      \
        LANEVAR(uint32_t, _q);                                                                \
        LANEVAR(int, _hok);                                                                   \
        LANEVAR(int, _inb);                                                                   \
        LANEVAR(int, _alive);                                                                 \
        LANEVAR(int, _pass);                                                                  \
        FOR_LANES                                                                             \
        {                                                                                     \
            LV(_hok) = (LV(ENT) >> 16) == sc.h;                                               \
            LV(_q) = (TPOS) + (LV(ENT) & ZD_TILE_MASK);                                       \
        }                                                                                     \
        const uint64_t _m_hash = BALLOT(_hok);                                                \
        if (!(_m_hash & 1ull)) {                                                              \
            (verdict) = 1; /* nothing (more) of this hash in this tile */                     \
            break;                                                                            \
        }                                                                                     \
        LANEVAR(int, _reach); /* near enough to be in the window (and so in the LDS ring) */    \
        FOR_LANES                                                                             \
        {                                                                                     \
            LV(_reach) = LV(_hok) && LV(_q) > st.base && sc.p - LV(_q) <= job.cfg.max_dist;   \
            LV(_inb) = LV(_reach) && MEMB(LV(_q));                                            \
        }                                                                                     \
        const uint64_t _m_reach = BALLOT(_reach);                                             \
        const uint64_t _m_in = BALLOT(_inb);                                                  \
        const int _head_lane = (!sc.head_seen && _m_in) ? CTZ64(_m_in) : 64;                  \
        FOR_LANES                                                                             \
        {                                                                                     \
            LV(_alive) = LV(_inb) && (sc.p - LV(_q) < job.cfg.max_dist || LANE == _head_lane); \
        }                                                                                     \
        if (job.cfg.hbits != 15u) {                                                           \
            /* another mem_level: the candidate at exactly MAX_DIST lives iff it heads the    \
             * reference's own chain, wherever it stands in this one */                       \
            LANEVAR(int, _edge);                                                              \
            FOR_LANES                                                                         \
            {                                                                                 \
                LV(_edge) = LV(_inb) && sc.p - LV(_q) == job.cfg.max_dist;                    \
            }                                                                                 \
            if (BALLOT(_edge) != 0) {                                                         \
                int _blk;                                                                     \
                LZ_HEAD_BLOCKED(sc.p - job.cfg.max_dist, sc.p, MEMB, _blk);                   \
                FOR_LANES                                                                     \
                {                                                                             \
                    if (LV(_edge))                                                            \
                        LV(_alive) = !_blk;                                                   \
                }                                                                             \
            }                                                                                 \
        }                                                                                     \
        const uint64_t _m_alive = BALLOT(_alive);                                             \
        if (!sc.head_seen) {                                                                  \
            if (_m_in == 0) {                                                                 \
                if (_m_hash & ~_m_reach)                                                      \
                    (verdict) = 3; /* whatever heads the chain is NIL or too far */           \
                else /* bucket entries, but none of them is in the chain: keep looking */     \
                    (verdict) = _m_hash != ~0ull ? 1 : 0;                                     \
                break;                                                                        \
            }                                                                                 \
            if (!((_m_alive >> _head_lane) & 1ull)) {                                         \
                (verdict) = 3; /* chain head is NIL or too far: longest_match not called */   \
                break;                                                                        \
            }                                                                                 \
            sc.head_seen = 1;                                                                 \
            if (sc.best >= sc.look) {                                                         \
                (verdict) = 2;                                                                \
                break;                                                                        \
            }                                                                                 \
            sc.sb = LZ_SPEEK32(sc.p + sc.best - 1) & 0xffffu;                                     \
        }                                                                                     \
        const int _ends = ((_m_hash & ~_m_reach) | (_m_in & ~_m_alive)) != 0; /* chain leaves the window */         \
        FOR_LANES                                                                             \
        {                                                                                     \
            int _c = 0;                                                                       \
            if (LV(_alive)) {                                                                 \
                _c = (lz_w32<L>(job, lds, st, (LV(_q))) & 0xffffu) == sc.s01 &&                  \
                     (lz_w32<L>(job, lds, st, (LV(_q) + sc.best - 1)) & 0xffffu) == sc.sb;       \
            }                                                                                 \
            LV(_pass) = _c;                                                                   \
        }                                                                                     \
        uint64_t _todo = BALLOT(_pass);                                                       \
        (verdict) = 0;                                                                        \
        while (_todo != 0) {                                                                  \
            const int _j = CTZ64(_todo);                                                      \
            const uint32_t _qj = READLANE(_q, _j);                                            \
            const uint32_t _len = lz_lcp<L>(job, lds, st, _qj, sc.p, sc.cap);                      \
            if (_len > sc.best) {                                                             \
                sc.where = _qj;                                                               \
                sc.best = _len;                                                               \
                if (_len >= sc.nice || --sc.budget == 0) {                                    \
                    (verdict) = 2;                                                            \
                    break;                                                                    \
                }                                                                             \
                sc.sb = LZ_SPEEK32(sc.p + sc.best - 1) & 0xffffu;                                 \
                FOR_LANES                                                                     \
                {                                                                             \
                    int _c = 0;                                                               \
                    if (LV(_alive) && LANE > _j) {                                            \
                        _c = (lz_w32<L>(job, lds, st, (LV(_q))) & 0xffffu) == sc.s01 &&          \
                             (lz_w32<L>(job, lds, st, (LV(_q) + sc.best - 1)) & 0xffffu) == sc.sb; \
                    }                                                                         \
                    LV(_pass) = _c;                                                           \
                }                                                                             \
                _todo = BALLOT(_pass);                                                        \
            } else {                                                                          \
                if (--sc.budget == 0) {                                                       \
                    (verdict) = 2;                                                            \
                    break;                                                                    \
                }                                                                             \
                _todo &= _todo - 1;                                                           \
            }                                                                                 \
        }                                                                                     \
        if ((verdict) == 0) {                                                                 \
            if (_ends)                                                                        \
                (verdict) = 2;                                                                \
            else if (_m_hash != ~0ull)                                                        \
                (verdict) = 1; /* bucket (or tile) exhausted */                               \
        }                                                                                     \
    } while (0)

/* Walk the rest of one run (batches after the first) in groups of four batches:
 * four independent coalesced loads are in flight together, so a long chain costs
 * one memory round trip per 256 candidates instead of one per 64. */
#define LZ_WALK_RUN(RUN, HI, TPOS, MEMB, verdict)                                                  \
    do {                                                                                      \
        int32_t _hi = (HI)-WAVE;                                                              \
        while ((verdict) == 0 && _hi >= 0) {                                                  \
            LANEVAR(uint32_t, _g0);                                                           \
            LANEVAR(uint32_t, _g1);                                                           \
            LANEVAR(uint32_t, _g2);                                                           \
            LANEVAR(uint32_t, _g3);                                                           \
            FOR_LANES                                                                         \
            {                                                                                 \
                int32_t _i = _hi - LANE;                                                      \
                LV(_g0) = _i >= 0 ? (RUN)[_i] : ZD_ENTRY_NONE;                                \
                LV(_g1) = _i >= 64 ? (RUN)[_i - 64] : ZD_ENTRY_NONE;                          \
                LV(_g2) = _i >= 128 ? (RUN)[_i - 128] : ZD_ENTRY_NONE;                        \
                LV(_g3) = _i >= 192 ? (RUN)[_i - 192] : ZD_ENTRY_NONE;                        \
            }                                                                                 \
            LZ_EVAL_BATCH(_g0, TPOS, MEMB, verdict);                                          \
            if ((verdict) == 0)                                                               \
                LZ_EVAL_BATCH(_g1, TPOS, MEMB, verdict);                                      \
            if ((verdict) == 0)                                                               \
                LZ_EVAL_BATCH(_g2, TPOS, MEMB, verdict);                                      \
            if ((verdict) == 0)                                                               \
                LZ_EVAL_BATCH(_g3, TPOS, MEMB, verdict);                                      \
            _hi -= 4 * WAVE;                                                                  \
        }                                                                                     \
        if ((verdict) == 0)                                                                   \
            (verdict) = 1; /* ran off the start of the tile */                                \
    } while (0)

/* deflate_slow, reference src/deflate.c:1989-2122, flush == Z_FINISH */
template <class L>
DEV void lz_parse_lazy(const LzJob &job, L *lds)
{
    LzState st;
    st.lo = st.hi = st.wrap_base = 0;
    st.base = 0;
    st.data_end = 0;
    st.nsyms = st.nstaged = 0;
    st.nblocks = st.blk_sym0 = st.blk_in0 = 0;
    st.pr_hi = 0;
    st.n = job.nsched ? job.n0 : job.n;
    st.si = 0;
    st.it = 0;
    lz_fold(job, st);

    uint32_t p = 0, cur_len = 2, cur_at = 0;
    int pending = 0; /* match_available */

    /* candidate lists fetched one search ahead: slot N for position p+1, slot J for
     * the position the parse jumps to when the pending match is emitted */
    LANEVAR(uint32_t, nA);
    LANEVAR(uint32_t, nB);
    LANEVAR(uint32_t, jA);
    LANEVAR(uint32_t, jB);
    LANEVAR(uint32_t, cA);
    LANEVAR(uint32_t, cB);
    uint32_t n_at = 0xffffffffu, j_at = 0xffffffffu;
    FOR_LANES { LV(nA) = LV(nB) = LV(jA) = LV(jB) = LV(cA) = LV(cB) = ZD_ENTRY_NONE; }
    const uint32_t last_owner = job.n >= 3 ? job.n - 3 : 0; /* last position that has a rank */

    for (;;) {
        st.it = p;
        uint32_t look = st.data_end - p;
        if (look < ZD_MIN_LOOKAHEAD) {
            lz_refill(job, st, p);
            look = st.data_end - p;
            if (look == 0) {
                if (st.si >= job.nsched)
                    break;
                /* a joint: src/deflate.c:2108-2113, then the next call carries on from here */
                if (pending) {
                    (void)lz_put<L>(job, lds, st, UNI(lz_w8<L>(job, lds, st, p - 1)));
                    pending = 0;
                }
                if (!lz_joint_at_end(job, st, p))
                    break; /* cannot happen: the host's joints end where the run ends */
                continue;
            }
        }
        lz_ensure<L>(job, lds, st, p);
        lz_ensure_ranks<L>(job, lds, st, p);

        const uint32_t prev_len = cur_len, prev_at = cur_at;
        const int searching = look >= 3 && prev_len < job.cfg.lazy;
        cur_len = 2;

        if (searching) {
            /* this position's first batches: prefetched one search ago, or fetched now */
            if (p == n_at) {
                FOR_LANES { LV(cA) = LV(nA); LV(cB) = LV(nB); }
            } else if (p == j_at) {
                FOR_LANES { LV(cA) = LV(jA); LV(cB) = LV(jB); }
            } else {
                LZ_FETCH_FIRST(p, cA, cB);
            }
        }
        /* issue the fetches for the two positions one of which is searched next:
         * p+1, and (when a match is pending) the jump target p-1+prev_len */
        n_at = j_at = 0xffffffffu;
        if (job.n >= 3 && p + 1 <= last_owner) {
            n_at = p + 1;
            LZ_FETCH_FIRST(n_at, nA, nB);
        }
        if (prev_len >= 3 && job.n >= 3 && p - 1 + prev_len <= last_owner) {
            j_at = p - 1 + prev_len;
            LZ_FETCH_FIRST(j_at, jA, jB);
        }

        if (searching) {
            LzSearch sc;
            sc.p = p;
            const uint32_t w0 = UNI(lz_w32<L>(job, lds, st, (p)));
            sc.h = (((w0 & 0xff) << 10) ^ (((w0 >> 8) & 0xff) << 5) ^ ((w0 >> 16) & 0xff)) & ZD_HASH_MASK;
            sc.s01 = w0 & 0xffff;
            sc.sb = 0;
            sc.look = look;
            sc.cap = look < 258u ? look : 258u;
            sc.nice = job.cfg.nice < look ? job.cfg.nice : look;
            sc.best = prev_len;
            sc.budget = prev_len >= job.cfg.good ? job.cfg.chain >> 2 : job.cfg.chain;
            sc.where = cur_at;
            sc.head_seen = 0;

            const uint32_t tile = p >> 15;
            const uint32_t *runA = job.sorted + (uint64_t)tile * ZD_TILE;
            int verdict = 0;
            /* own tile: first batch from the slot, the rest in groups of four */
            LZ_EVAL_BATCH(cA, tile << 15, LZ_MEMB_LAZY, verdict);
            if (verdict == 0)
                LZ_WALK_RUN(runA, (int32_t)UNI(lds->prank[p & (LZ_PR - 1)]) - 1, tile << 15, LZ_MEMB_LAZY, verdict);
            if (verdict == 1 && tile != 0) {
                /* older tile */
                verdict = 0;
                LZ_EVAL_BATCH(cB, (tile - 1) << 15, LZ_MEMB_LAZY, verdict);
                if (verdict == 0)
                    LZ_WALK_RUN(runA - ZD_TILE, (int32_t)(int16_t)UNI(lds->phib[p & (LZ_PR - 1)]),
                                (tile - 1) << 15, LZ_MEMB_LAZY, verdict);
            }
            /* verdict 3: the reference would not have called longest_match */
            if (verdict != 3 && sc.head_seen) {
                cur_at = sc.where;
                cur_len = sc.best < look ? sc.best : look;
                if (cur_len <= 5 &&
                    (job.strategy == 1 || (cur_len == 3 && p - cur_at > ZD_TOO_FAR)))
                    cur_len = 2;
            }
        }
        if (prev_len >= 3 && cur_len <= prev_len) {
            const int full = lz_put<L>(job, lds, st, ((p - 1 - prev_at) << 16) | (prev_len - 3));
            pending = 0;
            cur_len = 2;
            p += prev_len - 1;
            if (full) {
                const uint32_t old_n = lz_cut(job, st, p, 0, ZD_CUT_FULL);
                if (old_n != 0xffffffffu)
                    lz_mark_holes<L>(lds, st, old_n, p);
            }
        } else if (pending) {
            const uint32_t c = UNI(lz_w8<L>(job, lds, st, p - 1));
            uint32_t old_n = 0xffffffffu;
            if (lz_put<L>(job, lds, st, c))
                old_n = lz_cut(job, st, p, 0, ZD_CUT_FULL);
            p++;
            if (old_n != 0xffffffffu)
                lz_mark_holes<L>(lds, st, old_n, p);
        } else {
            pending = 1;
            p++;
        }
    }
    if (pending)
        (void)lz_put<L>(job, lds, st, UNI(lz_w8<L>(job, lds, st, p - 1)));
    lz_cut_end(job, st, p);
    if (st.nstaged)
        lz_flush_stage<L>(job, lds, st);
    ON_LANE0
    {
        job.out->nsyms = st.nsyms;
        job.out->nblocks = st.nblocks;
    }
}

/* mark position x as inserted into its hash chain (greedy parser) */
template <class L>
DEV void lz_mark_inserted(L *lds, const LzState &st, uint32_t x)
{
    const uint32_t r = lz_ridx<L>(st, x);
    ON_LANE0 { lds->ins[r >> 5] |= 1u << (r & 31u); }
    WAVE_SYNC();
}

/* deflate_fast, reference src/deflate.c:1886-1982 (levels 1-3), flush == Z_FINISH.
 * Same candidate machinery as the lazy parser; a candidate belongs to the chain only
 * if the parse inserted it (short matches index their interior, long ones do not). */
/* The bytes ahead of the parse in registers: lane l holds the dword at wc_at + 4 l, so that what a
 * loop top reads of its own string (the first four bytes, the two at best_len - 1, the literal) costs
 * two v_readlane and a shift instead of a trip to memory -- for the parser without an LDS window
 * that trip is a global load in the middle of every position's chain of dependent steps. */
#undef LZ_SPEEK32
#define LZ_SPEEK32(pos) lz_cached32<L>(job, lds, st, wc, wc_at, (pos))
template <class L, class V>
DEV uint32_t lz_cached32(const LzJob &job, const L *lds, const LzState &st, const V &wc, uint32_t wc_at, uint32_t pos)
{
    const uint32_t d = pos - wc_at;
    if (d <= 4u * (WAVE - 2u)) {
        const uint32_t lo = READLANE(wc, d >> 2), hi = READLANE(wc, (d >> 2) + 1u);
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (d & 3u)));
    }
    return UNI(lz_w32<L>(job, lds, st, pos));
}

template <class L>
DEV void lz_parse_greedy(const LzJob &job, L *lds)
{
    LzState st;
    st.lo = st.hi = st.wrap_base = 0;
    st.base = 0;
    st.data_end = 0;
    st.nsyms = st.nstaged = 0;
    st.nblocks = st.blk_sym0 = st.blk_in0 = 0;
    st.pr_hi = 0;
    st.n = job.nsched ? job.n0 : job.n;
    st.si = 0;
    st.it = 0;
    lz_fold(job, st);

    uint32_t p = 0, len = 0, at = 0;
    uint32_t owed = 0; /* s->insert: strings at the end of a section that wait for their third byte */
    /* first candidate batches of the position being searched (c) and of the next one (n), asked for
     * one search ahead as in the lazy parser: after a literal the parse goes to p + 1.  (Three
     * positions ahead in three register slots was slower, 864 against 749 ms on BASELINE config 3:
     * the compiler waits for all outstanding loads where one slot is used.) */
    LANEVAR(uint32_t, cA);
    LANEVAR(uint32_t, cB);
    LANEVAR(uint32_t, nA);
    LANEVAR(uint32_t, nB);
    uint32_t n_at = 0xffffffffu;
    LANEVAR(uint32_t, wc);
    FOR_LANES { LV(wc) = 0; }
    uint32_t wc_at = 0x80000000u; /* (no position of a buffer is within reach of it: buffers are shorter than 2^29) */
    for (;;) {
        st.it = p;
        uint32_t look = st.data_end - p;
        if (look < ZD_MIN_LOOKAHEAD) {
            lz_refill(job, st, p);
            look = st.data_end - p;
            if (owed && look + owed >= 3u) {
                /* fill_window, src/deflate.c:1591-1612: these enter the chains even where a
                 * long match had skipped them */
                lz_ensure<L>(job, lds, st, p);
                uint32_t str = p - owed;
                while (owed) {
                    lz_mark_inserted<L>(lds, st, str);
                    str++;
                    owed--;
                    if (look + owed < 3u)
                        break;
                }
            }
            if (look == 0) {
                if (!lz_joint_at_end(job, st, p))
                    break;
                owed = p < 2u ? p : 2u; /* :1975 */
                continue;
            }
        }
        lz_ensure<L>(job, lds, st, p);
        if (p - wc_at > 4u * (WAVE - 2u) - 8u) { /* (p and the few bytes behind it are always in reach) */
            wc_at = p & ~3u;
            FOR_LANES { LV(wc) = lz_w32<L>(job, lds, st, wc_at + 4u * (uint32_t)LANE); }
        }

        if (look >= 3) {
            lz_mark_inserted<L>(lds, st, p);
            LzSearch sc;
            sc.p = p;
            const uint32_t w0 = LZ_SPEEK32(p);
            sc.h = (((w0 & 0xff) << 10) ^ (((w0 >> 8) & 0xff) << 5) ^ ((w0 >> 16) & 0xff)) & ZD_HASH_MASK;
            sc.s01 = w0 & 0xffff;
            sc.sb = 0;
            sc.look = look;
            sc.cap = look < 258u ? look : 258u;
            sc.nice = job.cfg.nice < look ? job.cfg.nice : look;
            sc.best = 2; /* prev_length stays MIN_MATCH-1 in deflate_fast */
            sc.budget = 2 >= job.cfg.good ? job.cfg.chain >> 2 : job.cfg.chain;
            sc.where = at;
            sc.head_seen = 0;

            const uint32_t tile = p >> 15;
            const uint32_t *runA = job.sorted + (uint64_t)tile * ZD_TILE;
            if (st.pr_hi <= p)
                st.pr_hi = p & ~(uint32_t)(WAVE - 1); /* (a long match was jumped over) */
            while (st.pr_hi <= p || (st.pr_hi < job.ntot && st.pr_hi < p + (LZ_PR_FAST - WAVE))) {
                FOR_LANES
                {
                    const uint32_t x = st.pr_hi + (uint32_t)LANE;
                    if (x < job.ntot) {
                        lds->prank[x & (LZ_PR_FAST - 1)] = job.rank[x];
                        lds->phib[x & (LZ_PR_FAST - 1)] = job.hib[x];
                    }
                }
                WAVE_SYNC();
                st.pr_hi += WAVE;
            }
            const int32_t hiA = (int32_t)UNI(lds->prank[p & (LZ_PR_FAST - 1)]) - 1;
            const int32_t hiB = tile ? (int32_t)(int16_t)UNI(lds->phib[p & (LZ_PR_FAST - 1)]) : -1;
            if (p == n_at) {
                FOR_LANES { LV(cA) = LV(nA); LV(cB) = LV(nB); }
            } else {
                LZ_FETCH_FIRST(p, cA, cB);
            }
            n_at = 0xffffffffu;
            if (p + 4 <= st.n && p + 1 < st.pr_hi) { /* p + 1 owns a string and its rank is staged */
                n_at = p + 1;
                LZ_FETCH_FIRST(n_at, nA, nB);
            }
            int verdict = 0;
            LZ_EVAL_BATCH(cA, tile << 15, LZ_MEMB_INS, verdict);
            if (verdict == 0)
                LZ_WALK_RUN(runA, hiA, tile << 15, LZ_MEMB_INS, verdict);
            if (verdict == 1 && tile != 0) {
                verdict = 0;
                LZ_EVAL_BATCH(cB, (tile - 1) << 15, LZ_MEMB_INS, verdict);
                if (verdict == 0)
                    LZ_WALK_RUN(runA - ZD_TILE, hiB, (tile - 1) << 15, LZ_MEMB_INS, verdict);
            }
            if (verdict != 3 && sc.head_seen) {
                at = sc.where;
                len = sc.best < look ? sc.best : look;
            }
        }
        int full;
        if (len >= 3) {
            full = lz_put<L>(job, lds, st, ((p - at) << 16) | (len - 3));
            look -= len;
            if (len <= job.cfg.lazy /* max_insert_length */ && look >= 3) {
                for (len--; len != 0; len--) {
                    p++;
                    lz_mark_inserted<L>(lds, st, p);
                }
                p++;
            } else {
                p += len;
                len = 0;
            }
        } else {
            full = lz_put<L>(job, lds, st, LZ_SPEEK32(p) & 0xffu);
            p++;
        }
        if (full)
            (void)lz_cut(job, st, p, 0, ZD_CUT_FULL);
    }
    lz_cut_end(job, st, p);
    if (st.nstaged)
        lz_flush_stage<L>(job, lds, st);
    ON_LANE0
    {
        job.out->nsyms = st.nsyms;
        job.out->nblocks = st.nblocks;
    }
}

#undef LZ_SPEEK32
#define LZ_SPEEK32(pos) UNI(lz_w32<L>(job, lds, st, (pos)))

#endif
