/*
 * match_table.h -- kernel 1c (opt-in): longest_match for EVERY position, ahead of the parse.
 *
 * At levels 4-9 everything longest_match (reference src/deflate.c:1400-1518) reads is
 * independent of the parse -- the chains (every position is inserted, :2018,2069-2075), the
 * window base (a function of the position), lookahead and nice_match -- except prev_length,
 * the length it must beat.  And deflate_slow (:1989-2122) only ever calls it with two values
 * of prev_length at a position p: 2 (no match pending), or the length its own call at p-1
 * returned.  So one table entry per position can answer every call the parse makes:
 *
 *   r2[p]  longest_match(p, prev_length = 2), after the TOO_FAR / Z_FILTERED rule (:2038-2047);
 *          flagged MT_RLOK when it also answers for ANY longer prev_length: the walk from a
 *          longer prev_length ends at the same candidate, so the answer is "the same match if it
 *          is longer than prev_length, none otherwise" (see below for when that is certain).
 *
 * The parser (lz_parse_seg.h) turns runs of known entries into HOPS -- everything deflate_slow does
 * from one position with no match pending to the next such position follows from r2[x], r2[x+1],
 * ... -- and uses single entries where a hop is not known.
 *
 * The search is NOT the reference's walk.  In zsc a candidate that fails the pre-check costs
 * nothing -- no chain budget, no change of best_len (:1462-1469) -- so as long as the budget
 * does not run out the walk's result is a staircase: the nearest candidate longer than
 * prev_length, then the nearest one beyond it that is longer still, ... until nice_match is
 * reached or the window ends.  A candidate longer than `best` shares best+1 bytes with p, so it
 * is on the chain of EVERY trigram p+j, j <= best-2, shifted by j: each step walks the shortest
 * of a few of those chains instead of p's own (the chain of " th" has 1 200 entries where the
 * one of "e q" has 12).  On the positions the parse really searches, the Canterbury-like text
 * looks at 5 candidates per input byte this way where the reference's walk looks at 43, the
 * bitmap at 3 instead of 260 (tools/rare_stats.c).
 *
 * The budget (:1430-1432,1508-1512) is charged by candidates that pass the pre-check.  While
 * best_len <= 4 those are exactly the improving ones (bytes best-1, best, 0, 1 and -- same hash,
 * same first two bytes -- byte 2 are then bytes 0..best); from best_len 5 on a candidate can
 * pass by coincidence.  An entry is complete only when   steps + (entries of p's own chain that
 * lie between the first step to a level >= 5 and the last step) < max_chain,   i.e. when the
 * budget cannot have ended the walk before the last step; it is MT_RLOK only when   steps +
 * (ALL entries of p's chain newer than the last step's candidate) < max_chain / 4,   the smallest
 * budget and the earliest level >= 5 any longer prev_length can bring.  Everything else -- a chain
 * longer than MT_CAP entries, more than MT_STEPS steps, a compare longer than MT_LCP bytes -- is
 * left MT_INCOMPLETE and searched by the parser on demand.
 *
 * One workgroup per tile of 32 768 positions, the tile's window (the previous tile, the tile, the
 * lookahead) in LDS, a lane per position, straight-line code with fixed trip counts.  Measured
 * (DESIGN.md section 5d): what the kernel costs is about what its hops save, so it is off by
 * default (ZSC_HIP_TABLE); two earlier versions -- a per-lane loop without bounds, and lanes taking
 * positions off a queue through a state machine of phases -- executed 5-7 times the instructions
 * (profiles/r03_pmc_sq_table_queue_statemachine_text.txt).
 */
#ifndef ZSC_MATCH_TABLE_H
#define ZSC_MATCH_TABLE_H

#include "lz_parse_seg.h"
#ifdef ZSC_WAVE_EMU
#include <stdio.h>
#include <stdlib.h>
#endif

#ifndef MT_WAVES
#define MT_WAVES 16
#endif
#ifndef MT_STEPS
#define MT_STEPS 3u /* records a search follows before it is left to the parser */
#endif
#ifndef MT_LCP
#define MT_LCP 64u /* bytes a lane compares before the position is left to the parser (a run of equal bytes would
                      cost every one of its positions 258 of them) */
#endif
#ifndef MT_EARLY
#define MT_EARLY 1 /* give a position up as soon as its shortest chain is longer than MT_CAP */
#endif
#ifndef MT_CAP
#define MT_CAP 16u /* entries of one chain a search looks at (MtJob.cap, a multiple of 4) */
#endif

struct MtLds {
    static constexpr uint32_t SPAN = 2u * ZD_TILE + 288u;
    uint8_t win[SPAN + 16];
};

typedef struct {
    const uint8_t *in; /* this buffer */
    uint32_t n;
    uint32_t start;         /* absolute position of the tile's first byte */
    const uint32_t *sorted; /* tile 0 of this buffer */
    const uint16_t *rank, *hib;
    const uint32_t *cnt;
    uint32_t *r2; /* this buffer */
    ZdLevel cfg;  /* window_bits 15 / mem_level 8 only */
    uint32_t strategy;
    uint32_t cap; /* entries of one chain a search may look at before it is left to the parser */
} MtJob;

/* phase 0: the window of the tile into LDS; bytes before the buffer are never addressed,
 * bytes behind it read as zero (fill_window, src/deflate.c:1616-1649) */
DEV void mt_phase_load(const MtJob &job, MtLds *lds, int w)
{
    const uint32_t wbase = job.start - ZD_TILE; /* wraps below zero for tile 0: only differences are used */
    const uint32_t first = job.start >= ZD_TILE ? 0u : ZD_TILE;
    for (uint32_t o = first + (uint32_t)w * WAVE * 16u; o < MtLds::SPAN + 16u; o += MT_WAVES * WAVE * 16u) {
        FOR_LANES
        {
            const uint32_t off = o + (uint32_t)LANE * 16u;
            if (off < MtLds::SPAN + 16u) {
                const uint32_t a = wbase + off;
                uint8_t *dst = &lds->win[off];
                if ((uint64_t)a + 16u <= job.n) {
                    COPY16(dst, job.in + a);
                } else {
                    for (uint32_t j = 0; j < 16; j++)
                        dst[j] = a + j < job.n ? job.in[a + j] : (uint8_t)0;
                }
            }
        }
    }
}

/* longest common prefix of the strings at window indices iq and ip, at most cap bytes */
DEV uint32_t mt_lcp(const uint8_t *win, uint32_t iq, uint32_t ip, uint32_t cap)
{
    uint32_t l = 0;
    while (l < cap) {
        const uint32_t x = lds_u32(win, iq + l) ^ lds_u32(win, ip + l);
        if (x != 0) {
            l += (uint32_t)CTZ32(x) >> 3;
            break;
        }
        l += 4;
    }
    return l < cap ? l : cap;
}

/* entry number i of the sorted array the chain of a position lies in (its own tile's, continued
 * downwards by the previous tile's, which precedes it in memory): the tile-relative position it
 * names.  Chain entry v of a position with rank rk, link hb and nA earlier members in its own tile
 * is number rk - 1 - v for v < nA, and hb - ZD_TILE - (v - nA) after that. */
DEV int32_t mt_slot(uint32_t rk, uint32_t hb, uint32_t nA, uint32_t v)
{
    return v < nA ? (int32_t)(rk - 1u - v) : (int32_t)(hb - (v - nA)) - (int32_t)ZD_TILE;
}

/* longest_match(p, prev_length = 2) as a table entry; base0 = window base at the tile's start.
 * Straight-line code with fixed trip counts: what does not fit them is left incomplete. */
DEV uint32_t mt_search(const MtJob &job, const MtLds *lds, uint32_t p, uint32_t base0)
{
    const uint32_t n = job.n;
    if ((uint64_t)p + 3u > n)
        return MT_NONE | MT_RLOK; /* :2027: fewer than three bytes ahead */
    const uint32_t look = n - p; /* what matters of it: fill_window keeps MIN_LOOKAHEAD bytes ahead until the input ends */
    const uint32_t cn = job.cnt[p];
    const uint32_t nA = cn & 0xffffu, total = nA + (cn >> 16);
    if (total == 0)
        return MT_NONE | MT_RLOK;
    const uint32_t rk = job.rank[p], hb = job.hib[p];
    const uint32_t tileA = p & ~ZD_TILE_MASK;
    uint32_t base = base0;
    for (int k = 0; k < 2; k++) { /* sg_base_at, from the tile's base on: at most two slides inside a tile */
        const uint64_t end = (uint64_t)base + 2ull * ZD_TILE;
        const uint32_t data_end = end < n ? (uint32_t)end : n;
        const int slide = (uint64_t)p + ZD_MIN_LOOKAHEAD > data_end && p - base >= ZD_TILE + ZD_MAX_DIST;
        base += slide ? ZD_TILE : 0u;
    }
    const uint32_t floor_pos = p - base > ZD_MAX_DIST ? p - ZD_MAX_DIST : base;
    const uint32_t *runA = job.sorted + (uint64_t)(p >> 15) * ZD_TILE;
    /* the chain head may lie at exactly MAX_DIST (:2027-2028), later links may not (:1512) */
    const uint32_t q0 = tileA + (runA[mt_slot(rk, hb, nA, 0u)] & ZD_TILE_MASK) - (nA ? 0u : ZD_TILE);
    if (!(q0 > base && p - q0 <= ZD_MAX_DIST))
        return MT_NONE | MT_RLOK; /* longest_match is not called */
    const uint32_t cap = look < 258u ? look : 258u;
    const uint32_t nice = job.cfg.nice < look ? job.cfg.nice : look;
    const uint32_t wbase = job.start - ZD_TILE;
    const uint8_t *win = lds->win;
    const uint32_t ip = p - wbase;

    uint32_t b = 2u, bnd = p, where = 0, nrec = 0;
    uint32_t qs = 0xffffffffu; /* where the walk first stands at a level >= 5 */
    int open = 1;              /* the walk is not known to be over */
    for (uint32_t step = 0; step < MT_STEPS && open; step++) {
        /* the shortest of the chains of p, p+1, p+b-3, p+b-2 */
        uint32_t j = 0, cj = cn, tj = total, rkj = rk, hbj = hb;
        if (b >= 3u) {
            const uint32_t jm = b - 2u;
            const uint32_t o1 = 1u, o2 = jm >= 2u ? jm - 1u : 1u, o3 = jm;
            const int ok1 = (uint64_t)p + o1 + 3u <= n, ok2 = (uint64_t)p + o2 + 3u <= n, ok3 = (uint64_t)p + o3 + 3u <= n;
            const uint32_t c1 = job.cnt[ok1 ? p + o1 : p], c2 = job.cnt[ok2 ? p + o2 : p], c3 = job.cnt[ok3 ? p + o3 : p];
            const uint32_t t1 = ok1 ? (c1 & 0xffffu) + (c1 >> 16) : 0xffffffffu, t2 = ok2 ? (c2 & 0xffffu) + (c2 >> 16) : 0xffffffffu,
                           t3 = ok3 ? (c3 & 0xffffu) + (c3 >> 16) : 0xffffffffu;
            if (t1 < tj) {
                tj = t1, cj = c1, j = o1;
            }
            if (t2 < tj) {
                tj = t2, cj = c2, j = o2;
            }
            if (t3 < tj) {
                tj = t3, cj = c3, j = o3;
            }
            if (j) {
                rkj = job.rank[p + j];
                hbj = job.hib[p + j];
            }
        }
        const uint32_t tpos = (p + j) & ~ZD_TILE_MASK;
        const uint32_t *run = job.sorted + (uint64_t)((p + j) >> 15) * ZD_TILE;
        const uint32_t nAj = cj & 0xffffu;
        const uint32_t sb = (uint32_t)win[ip + b] << 8 | win[ip + b - 1u];
        int found = 0, ended = 0;
        uint32_t fq = 0, flen = 0;
        open = 0;
        if (b >= 3u && tj > job.cap && MT_EARLY)
            return MT_INCOMPLETE; /* a chain longer than a lane walks: not worth starting on */
        const uint32_t lim = tj < job.cap ? tj : job.cap;
        uint32_t v = 0;
        /* four entries a trip (asked for together), each looked at with as few branches as it takes:
         * a wave pays for every branch of every lane */
        while (v < lim && !(ended | found)) {
            uint32_t qq[4];
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t vk = v + k < tj ? v + k : tj - 1u;
                qq[k] = tpos + (run[mt_slot(rkj, hbj, nAj, vk)] & ZD_TILE_MASK) - (vk < nAj ? 0u : ZD_TILE);
            }
            for (uint32_t k = 0; k < 4u; k++) {
                const uint32_t qj = qq[k], q = qj - j;
                const int used = (ended | found) || v + k >= lim;
                /* the chain leaves the buffer or the window (:1512) */
                const int gone = !used && (qj < j || (q < bnd && !(q > floor_pos || q == q0)));
                /* (an entry newer than the last record was seen at an earlier level) */
                const int cand = !used && !gone && q < bnd;
                const uint32_t iq = cand ? q - wbase : ip;
                const uint32_t two = (uint32_t)win[iq + b] << 8 | win[iq + b - 1u];
                if (cand && two == sb) {
                    const uint32_t len = mt_lcp(win, iq, ip, cap < MT_LCP ? cap : MT_LCP);
                    if (len >= MT_LCP && len < cap)
                        return MT_INCOMPLETE; /* a long one: the parser's wave compares 256 bytes a step */
                    if (len > b) {
                        found = 1;
                        fq = q;
                        flen = len;
                    }
                }
                ended |= gone;
            }
            v += 4u;
        }
        if (found) {
            nrec++;
            b = flen;
            where = fq;
            bnd = fq;
            open = b < nice;
            if (open && qs == 0xffffffffu && b >= 5u)
                qs = fq;
        } else if (!ended && lim < tj) {
            return MT_INCOMPLETE; /* more of this chain than a lane looks at */
        }
    }
    if (open)
        return MT_INCOMPLETE; /* more records than a lane follows */
    uint32_t res = MT_NONE | MT_RLOK;
    if (nrec != 0) {
        /* Could the chain budget have ended the walk before its last record?  It is charged by the
         * records, and by candidates that pass the pre-check without being longer: none while
         * best_len <= 4 (see the top of the file), so at most as many as p's own chain has entries
         * between the first record at a level >= 5 and the last record. */
        uint32_t newer = 0, between = 0;
        if (where != q0) {
            const uint32_t r1 = job.rank[where];
            newer = (where >> 15) == (p >> 15) ? rk - 1u - r1 : nA + (hb - r1);
            if (qs != 0xffffffffu && qs != where) {
                const uint32_t r0 = job.rank[qs];
                between = newer - ((qs >> 15) == (p >> 15) ? rk - 1u - r0 : nA + (hb - r0)) - 1u;
            }
        }
        uint32_t len = b < look ? b : look;
        if (len <= 5u) { /* :2038-2047 */
            if (job.strategy == 1u)
                len = 2;
            else if (len == 3u && p - where > ZD_TOO_FAR)
                len = 2;
        }
        res = len <= 2u ? MT_NONE : MT_PACK(len, p - where);
        if (nrec + between >= job.cfg.chain)
            res = MT_INCOMPLETE;
        else if (nrec + newer < ((uint32_t)job.cfg.chain >> 2))
            res |= MT_RLOK; /* the walk from any longer prev_length ends at the same record, whatever passes on its way */
    }
    return res;
}

/* the tile's entries, a lane per position */
DEV void mt_phase_search(const MtJob &job, const MtLds *lds, int w, uint32_t base0)
{
    const uint32_t left = job.n - job.start;
    const uint32_t m = left < ZD_TILE ? left : ZD_TILE;
    for (uint32_t i0 = (uint32_t)w * WAVE; i0 < m; i0 += MT_WAVES * WAVE) {
        FOR_LANES
        {
            const uint32_t i = i0 + (uint32_t)LANE;
            if (i < m)
                job.r2[job.start + i] = mt_search(job, lds, job.start + i, base0);
        }
    }
}

#endif
