/*
 * match_table.h -- kernel 1c: longest_match for EVERY position, ahead of the parse.
 *
 * At levels 4-9 everything longest_match (reference src/deflate.c:1400-1518) reads is
 * independent of the parse -- the chains (every position is inserted, :2018,2069-2075), the
 * window base (a function of the position), lookahead and nice_match -- except prev_length,
 * the length it must beat.  And deflate_slow (:1989-2122) only ever calls it with two values
 * of prev_length at a position p: 2 (no match pending), or the length its own call at p-1
 * returned.  So a table with two entries per position answers every call the parse makes:
 *
 *   r2[p]  longest_match(p, prev_length = 2), after the TOO_FAR / Z_FILTERED rule (:2038-2047)
 *   rl[p]  longest_match(p, prev_length = length in r2[p-1]) where that is 3 .. max_lazy-1
 *
 * (The call at p-1 may itself have been made with a pending match; if the chain budget ended
 * that walk elsewhere than the walk from 2, its result is not the key rl[p] was made for.  The
 * parser checks the key and searches itself in that case: lz_parse_seg.h.)
 *
 * The search is NOT the reference's walk.  In zsc a candidate that fails the pre-check costs
 * nothing -- no chain budget, no change of best_len (:1462-1469) -- so as long as the budget
 * does not run out the walk's result is a staircase: the nearest candidate longer than
 * prev_length, then the nearest one beyond it that is longer still, ... until nice_match is
 * reached or the window ends.  A candidate longer than `best` shares best+1 bytes with p, so it
 * is on the chain of EVERY trigram p+j, j <= best-2, shifted by j: each step walks the shortest
 * of a few of those chains instead of p's own (the chain of " th" has 1 200 entries where the
 * one of "e q" has 12).  On the Canterbury-like text this looks at 5 candidates per input byte
 * where the reference's walk looks at 43, on the bitmap 3 instead of 260.
 *
 * The budget (:1430-1432,1508-1512) is charged by candidates that pass the pre-check.  While
 * best_len <= 4 those are exactly the improving ones (bytes best-1, best, 0, 1 and -- same hash,
 * same first two bytes -- byte 2 are then bytes 0..best); from best_len 5 on a candidate can
 * pass by coincidence.  An entry is only written when   records + (candidates of p's chain that
 * lie between the first record reached at a level >= 5 and the last record) < budget,   i.e.
 * when the budget cannot have ended the walk before the last record.  Everything else -- that
 * bound exceeded, or more than MT_CAP candidates looked at -- is left MT_INCOMPLETE and
 * searched by the parser on demand, with the walk it always had.
 *
 * One workgroup per tile of 32 768 positions, the tile's window (the previous tile, the tile,
 * 512 bytes of lookahead) in LDS, a lane per position.
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
#ifndef MT_CAP
#define MT_CAP 48u /* entries a search looks at before it is left to the parser (MtJob.cap) */
#endif
#ifndef MT_WALK
#define MT_WALK 8u /* entries a lane looks at per round */
#endif
#define MT_CHUNK ((uint32_t)MT_WAVES * WAVE) /* positions per step of the workgroup */
#define MT_META (MT_CHUNK + 264u)             /* positions whose chain records are staged: one before the chunk, 258 + slack behind */
#define MT_KEY_NONE 0xffffu

struct MtLds {
    static constexpr uint32_t SPAN = 2u * ZD_TILE + 288u;
    uint8_t win[SPAN + 16];
    uint32_t mcnt[MT_META];        /* cnt[] of the positions from one before the chunk on */
    uint32_t mrh[MT_META];         /* rank | hib << 16 of the same */
    uint16_t key[2][MT_CHUNK + 1]; /* length in r2 of the chunk's positions (slot 1 + i), double-buffered */
    uint32_t queue[2];             /* next position of the chunk to search: r2, rl */
};

typedef struct {
    const uint8_t *in; /* this buffer */
    uint32_t n;
    uint32_t start;         /* absolute position of the tile's first byte */
    const uint32_t *sorted; /* tile 0 of this buffer */
    const uint16_t *rank, *hib;
    const uint32_t *cnt;
    uint32_t *r2, *rl; /* this buffer */
    ZdLevel cfg;       /* window_bits 15 / mem_level 8 only */
    uint32_t strategy;
    uint32_t cap; /* entries a search may look at before it is left to the parser */
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

/* before chunk c: the chain records of its positions (and of what a search can ask for around
 * them) into LDS, the two work queues back to their start */
DEV void mt_phase_stage(const MtJob &job, MtLds *lds, int w, uint32_t c)
{
    const uint32_t m0 = job.start + c * MT_CHUNK - 1u; /* (wraps for the buffer's first chunk: slot 0 is not used then) */
    for (uint32_t i0 = (uint32_t)w * WAVE; i0 < MT_META; i0 += MT_WAVES * WAVE) {
        FOR_LANES
        {
            const uint32_t i = i0 + (uint32_t)LANE;
            if (i < MT_META) {
                const uint32_t x = m0 + i;
                const int owner = !(i == 0 && m0 == 0xffffffffu) && (uint64_t)x + 3u <= job.n;
                lds->mcnt[i] = owner ? job.cnt[x] : 0u;
                lds->mrh[i] = owner ? (uint32_t)job.rank[x] | ((uint32_t)job.hib[x] << 16) : 0u;
            }
        }
    }
    if (w == 0) {
        FOR_LANES
        {
            if (LANE < 2)
                lds->queue[LANE] = 0;
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

/* entry v (0 = newest) of the chain of a position with rank rk, link hb and nA earlier members in
 * its own tile, which starts at tpos: the absolute position it names */
DEV uint32_t mt_entry(const MtJob &job, uint32_t tpos, uint32_t rk, uint32_t hb, uint32_t nA, uint32_t v)
{
    const uint32_t *run = job.sorted + (uint64_t)(tpos >> 15) * ZD_TILE;
    if (v < nA)
        return tpos + (run[rk - 1u - v] & ZD_TILE_MASK);
    return tpos - ZD_TILE + ((run - ZD_TILE)[hb - (v - nA)] & ZD_TILE_MASK);
}

/* four entries at once, v .. v+3 (one 16-byte load where the four lie in one run of one tile);
 * entries past the end of the chain come back as 0xffffffff */
typedef struct {
    uint32_t q0, q1, q2, q3;
} MtBlock;
DEV MtBlock mt_block(const MtJob &job, uint32_t tpos, uint32_t rk, uint32_t hb, uint32_t nA, uint32_t tot, uint32_t v)
{
    MtBlock o;
    const uint32_t *run = job.sorted + (uint64_t)(tpos >> 15) * ZD_TILE;
    if (v + 4u <= nA) {
        const uint32_t *e = run + (rk - 4u - v);
        const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
        o.q0 = tpos + (e3 & ZD_TILE_MASK);
        o.q1 = tpos + (e2 & ZD_TILE_MASK);
        o.q2 = tpos + (e1 & ZD_TILE_MASK);
        o.q3 = tpos + (e0 & ZD_TILE_MASK);
    } else if (v >= nA && v + 4u <= tot) {
        const uint32_t *e = run - ZD_TILE + (hb - (v - nA) - 3u);
        const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
        o.q0 = tpos - ZD_TILE + (e3 & ZD_TILE_MASK);
        o.q1 = tpos - ZD_TILE + (e2 & ZD_TILE_MASK);
        o.q2 = tpos - ZD_TILE + (e1 & ZD_TILE_MASK);
        o.q3 = tpos - ZD_TILE + (e0 & ZD_TILE_MASK);
    } else {
        o.q0 = v < tot ? mt_entry(job, tpos, rk, hb, nA, v) : 0xffffffffu;
        o.q1 = v + 1u < tot ? mt_entry(job, tpos, rk, hb, nA, v + 1u) : 0xffffffffu;
        o.q2 = v + 2u < tot ? mt_entry(job, tpos, rk, hb, nA, v + 2u) : 0xffffffffu;
        o.q3 = v + 3u < tot ? mt_entry(job, tpos, rk, hb, nA, v + 3u) : 0xffffffffu;
    }
    return o;
}

/* number of entries of p's chain that are newer than q (q on the chain) */
DEV uint32_t mt_index(const MtJob &job, uint32_t p, uint32_t rk, uint32_t hb, uint32_t nA, uint32_t q)
{
    const uint32_t rq = job.rank[q];
    return (q >> 15) == (p >> 15) ? rk - 1u - rq : nA + (hb - rq);
}

/* ---- the searches of one chunk, a lane per search ------------------------------------------
 *
 * Searches differ in length by two orders of magnitude, and what they do at any moment differs
 * too: taking a position, choosing a chain, walking it, comparing a candidate that passed the
 * two-byte test, wrapping up.  Run as one loop per lane the wave would execute, in EVERY trip,
 * the most expensive thing any of its lanes wants.  So the lanes take positions off a queue --
 * one that is done with a short search starts the next -- and a round of the wave is a fixed
 * sequence of blocks, each executed once for all the lanes that want it.  The blocks are ordered
 * so that nothing a block loads from global memory is used before the NEXT round (walk, compare,
 * wrap up, new position, choose chain): a round costs its instructions, not its round trips. */
#define MT_S_NEW 0u
#define MT_S_PICK 1u
#define MT_S_WALK 2u
#define MT_S_LCP 3u
#define MT_S_FIN 4u  /* the search is over: ask for what the budget test needs */
#define MT_S_FIN2 5u /* write the entry */
#define MT_S_IDLE 6u

#define MT_K_INC 0x8000u  /* key slot: the position's r2 is incomplete */
#define MT_K_RLOK 0x4000u /* key slot: the position's r2 answers for longer prev_lengths too */

typedef struct {
    LANEVAR(uint32_t, st);
    LANEVAR(uint32_t, idx);   /* the item: position start + c * MT_CHUNK + idx (idx == MT_CHUNK: the position before the tile) */
    LANEVAR(uint32_t, p);
    LANEVAR(uint32_t, b0);
    LANEVAR(uint32_t, res);   /* the entry, once known */
    LANEVAR(uint32_t, q0);    /* head of p's chain: the entry as loaded while hd is set, the position after */
    LANEVAR(uint32_t, hd);    /* the head has not been looked at yet */
    LANEVAR(uint32_t, base);
    LANEVAR(uint32_t, b);
    LANEVAR(uint32_t, bnd);
    LANEVAR(uint32_t, where);
    LANEVAR(uint32_t, nrec);
    LANEVAR(uint32_t, qs);    /* where the walk first stands at a level >= 5; in MT_S_FIN2: its rank */
    LANEVAR(uint32_t, looked);
    LANEVAR(uint32_t, j);     /* the chain being walked: that of p + j */
    LANEVAR(uint32_t, rhj);   /* its rank | hib << 16 */
    LANEVAR(uint32_t, cj);    /* its cnt */
    LANEVAR(uint32_t, v);     /* next entry */
    LANEVAR(uint32_t, sb);    /* the two bytes at p + b - 1 */
    LANEVAR(uint32_t, qp);    /* the candidate to compare; in MT_S_FIN2: rank of the last record */
    LANEVAR(MtBlock, cur);
    LANEVAR(MtBlock, nxt);
} MtWave;

/* an entry to the table (and, for r2, its length and flags to the chunk's keys) */
DEV void mt_store(const MtJob &job, MtLds *lds, uint32_t kind, uint32_t c, uint32_t npos, uint32_t idx, uint32_t p,
                  uint32_t res)
{
    if (kind == 0u) {
        if (idx < npos)
            job.r2[p] = res;
        lds->key[idx < npos ? (c & 1u) : 1u][idx < npos ? 1u + idx : MT_CHUNK] =
            (uint16_t)((res & MT_INCOMPLETE) ? MT_K_INC : (MT_LEN(res) | ((res & MT_RLOK) ? MT_K_RLOK : 0u)));
    } else {
        job.rl[p] = res & ~MT_RLOK;
    }
}

/* kind 0: r2 of the chunk's positions (and, in the tile's first chunk, of the position before the
 * tile, for its key); kind 1: rl, keyed by the length in r2 of the position before, where r2 of the
 * position itself does not answer for it */
DEV void mt_phase_search(const MtJob &job, MtLds *lds, int w, uint32_t c, uint32_t kind, uint32_t base0)
{
    (void)w;
    const uint32_t n = job.n;
    const uint32_t c0 = job.start + c * MT_CHUNK;
    const uint32_t left = n - c0 < job.start + ZD_TILE - c0 ? n - c0 : job.start + ZD_TILE - c0;
    const uint32_t npos = left < MT_CHUNK ? left : MT_CHUNK;
    const uint32_t nitems = npos + ((kind == 0u && c == 0u && job.start != 0u) ? 1u : 0u);
    const uint32_t wbase = job.start - ZD_TILE;
    const uint8_t *win = lds->win;
    MtWave ws;
    FOR_LANES { LV(ws.st) = MT_S_NEW; }
    for (;;) {
        /* ---- up to MT_WALK entries of the chain (asked for in an earlier round) ---- */
        for (uint32_t t = 0; t < MT_WALK; t++) {
            FOR_LANES
            {
                if (LV(ws.st) == MT_S_WALK) {
                    const uint32_t p = LV(ws.p);
                    if (LV(ws.hd)) {
                        /* the chain head may lie at exactly MAX_DIST (:2027-2028), later links may not (:1512) */
                        const uint32_t mi = p - (c0 - 1u);
                        const uint32_t nA = lds->mcnt[mi] & 0xffffu;
                        const uint32_t q0 = (p & ~ZD_TILE_MASK) - (nA ? 0u : ZD_TILE) + (LV(ws.q0) & ZD_TILE_MASK);
                        LV(ws.q0) = q0;
                        LV(ws.hd) = 0;
                        if (!(q0 > LV(ws.base) && p - q0 <= ZD_MAX_DIST))
                            LV(ws.st) = MT_S_FIN; /* longest_match is not called: the entry stays MT_NONE */
                    }
                }
                if (LV(ws.st) == MT_S_WALK) {
                    const uint32_t cj = LV(ws.cj), tj = (cj & 0xffffu) + (cj >> 16), v = LV(ws.v), j = LV(ws.j);
                    if (v >= tj) {
                        LV(ws.st) = MT_S_FIN; /* no candidate longer than b */
                    } else {
                        const uint32_t p = LV(ws.p);
                        const uint32_t k = v & 3u;
                        const uint32_t qj = k == 0u ? LV(ws.cur).q0 : k == 1u ? LV(ws.cur).q1 : k == 2u ? LV(ws.cur).q2 : LV(ws.cur).q3;
                        LV(ws.v) = v + 1u;
                        if (k == 3u) {
                            /* the next four are here; the four after them on their way */
                            LV(ws.cur) = LV(ws.nxt);
                            if (v + 5u < tj) {
                                const uint32_t rhj = LV(ws.rhj);
                                LV(ws.nxt) = mt_block(job, (p + j) & ~ZD_TILE_MASK, rhj & 0xffffu, rhj >> 16,
                                                      cj & 0xffffu, tj, v + 5u);
                            }
                        }
                        const uint32_t q = qj - j;
                        const uint32_t base = LV(ws.base);
                        const uint32_t floor_pos = p - base > ZD_MAX_DIST ? p - ZD_MAX_DIST : base;
                        if (qj < j) {
                            LV(ws.st) = MT_S_FIN;
                        } else if (++LV(ws.looked) > job.cap) {
                            LV(ws.res) = MT_INCOMPLETE;
                            LV(ws.nrec) = 0;
                            LV(ws.st) = MT_S_FIN;
                        } else if (q >= LV(ws.bnd)) {
                            /* newer than the last record: seen at an earlier level */
                        } else if (!(q > floor_pos || q == LV(ws.q0))) {
                            LV(ws.st) = MT_S_FIN; /* the chain leaves the window (:1512) */
                        } else {
                            const uint32_t iq = q - wbase, b = LV(ws.b);
                            if (((uint32_t)win[iq + b] << 8 | win[iq + b - 1u]) == LV(ws.sb)) {
                                LV(ws.qp) = q;
                                LV(ws.st) = MT_S_LCP;
                            }
                        }
                    }
                }
            }
        }
        /* ---- the candidate that showed the two bytes: is it longer? ---- */
        FOR_LANES
        {
            if (LV(ws.st) == MT_S_LCP) {
                const uint32_t p = LV(ws.p), q = LV(ws.qp), look = n - p;
                const uint32_t cap = look < 258u ? look : 258u;
                const uint32_t len = mt_lcp(win, q - wbase, p - wbase, cap);
                LV(ws.st) = MT_S_WALK;
                if (len > LV(ws.b)) {
                    const uint32_t nice = job.cfg.nice < look ? job.cfg.nice : look;
                    LV(ws.nrec)++;
                    LV(ws.b) = len;
                    LV(ws.where) = q;
                    LV(ws.bnd) = q;
                    LV(ws.st) = len >= nice ? MT_S_FIN : MT_S_PICK;
                    if (LV(ws.qs) == 0xffffffffu && len >= 5u && len < nice)
                        LV(ws.qs) = q;
                }
            }
        }
        /* ---- the entry: written with what the block below asked for in the round before ---- */
        FOR_LANES
        {
            if (LV(ws.st) == MT_S_FIN2) {
                const uint32_t p = LV(ws.p), b0 = LV(ws.b0), idx = LV(ws.idx);
                uint32_t res = LV(ws.res);
                if (LV(ws.nrec) != 0) {
                    const uint32_t where = LV(ws.where), look = n - p, b = LV(ws.b);
                    /* Could the chain budget have ended the walk before its last record?  It is charged
                     * by the records, and by candidates that pass the pre-check without being longer:
                     * none while best_len <= 4 (see the top of the file), so at most as many as p's own
                     * chain has entries between the first record at a level >= 5 and the last record. */
                    const uint32_t mi = p - (c0 - 1u);
                    const uint32_t cn = lds->mcnt[mi], rh = lds->mrh[mi];
                    const uint32_t qs = LV(ws.sb); /* (the position; its rank is in ws.qs) */
                    const uint32_t r1 = LV(ws.qp), r0 = LV(ws.qs);
                    const uint32_t newer = (where >> 15) == (p >> 15) ? (rh & 0xffffu) - 1u - r1 : (cn & 0xffffu) + ((rh >> 16) - r1);
                    uint32_t between = 0;
                    if (qs == p)
                        between = newer;
                    else if (qs != 0xffffffffu && qs != where)
                        between = newer - ((qs >> 15) == (p >> 15) ? (rh & 0xffffu) - 1u - r0 : (cn & 0xffffu) + ((rh >> 16) - r0)) - 1u;
                    const uint32_t charged = LV(ws.nrec) + between;
                    const uint32_t budget = b0 >= job.cfg.good ? (uint32_t)job.cfg.chain >> 2 : job.cfg.chain;
                    uint32_t len = b < look ? b : look;
                    if (len <= 5u) { /* :2038-2047 */
                        if (job.strategy == 1u)
                            len = 2;
                        else if (len == 3u && p - where > ZD_TOO_FAR)
                            len = 2;
                    }
                    res = len <= b0 ? MT_NONE : MT_PACK(len, p - where);
                    if (charged >= budget)
                        res = MT_INCOMPLETE;
                    else if (LV(ws.nrec) + newer < ((uint32_t)job.cfg.chain >> 2))
                        res |= MT_RLOK; /* the walk from any longer prev_length ends at the same record, whatever passes on its way */
                } else if (!(res & MT_INCOMPLETE)) {
                    res |= MT_RLOK; /* no match at all: none longer than anything either */
                }
                mt_store(job, lds, kind, c, npos, idx, p, res);
                LV(ws.st) = MT_S_NEW;
            }
        }
        FOR_LANES
        {
            if (LV(ws.st) == MT_S_FIN) {
                if (LV(ws.nrec) != 0) {
                    const uint32_t qs = LV(ws.qs), p = LV(ws.p);
                    LV(ws.qp) = job.rank[LV(ws.where)];
                    LV(ws.sb) = qs;
                    if (qs != 0xffffffffu && qs != p)
                        LV(ws.qs) = job.rank[qs];
                }
                LV(ws.st) = MT_S_FIN2;
            }
        }
        /* ---- a position off the queue ---- */
        FOR_LANES
        {
            while (LV(ws.st) == MT_S_NEW) {
                const uint32_t idx = LDS_FETCH_ADD_U32(&lds->queue[kind], 1u);
                if (idx >= nitems) {
                    LV(ws.st) = MT_S_IDLE;
                    break;
                }
                const uint32_t p = idx < npos ? c0 + idx : job.start - 1u;
                uint32_t b0 = 2u, res = MT_NONE | MT_RLOK;
                int search = 1;
                if (kind != 0u) {
                    const uint32_t key = idx ? lds->key[c & 1u][idx] : lds->key[(c & 1u) ^ 1u][MT_CHUNK];
                    const uint32_t own = lds->key[c & 1u][1u + idx];
                    b0 = key & 0x1ffu;
                    if (own & MT_K_RLOK)
                        continue; /* r2[p] answers: rl[p] is not read */
                    if (key & MT_K_INC) {
                        res = MT_INCOMPLETE;
                        search = 0;
                    } else if (b0 < 3u) {
                        search = 0;
                    }
                }
                /* :2027: three bytes ahead, and the previous match not good enough already */
                if ((uint64_t)p + 3u > n || b0 >= job.cfg.lazy)
                    search = 0;
                if (search) {
                    const uint32_t mi = p - (c0 - 1u);
                    const uint32_t cn = lds->mcnt[mi], rh = lds->mrh[mi];
                    const uint32_t nA = cn & 0xffffu, total = nA + (cn >> 16);
                    const uint32_t look = n - p; /* what matters of it: fill_window keeps MIN_LOOKAHEAD bytes ahead until the input ends */
                    search = total != 0 && b0 < look;
                    if (search) {
                        uint32_t base = idx < npos ? base0 : (base0 >= ZD_TILE ? base0 - ZD_TILE : 0u);
                        for (;;) { /* sg_base_at, from the tile's base on (at most two steps) */
                            const uint64_t end = (uint64_t)base + 2ull * ZD_TILE;
                            const uint32_t data_end = end < n ? (uint32_t)end : n;
                            if ((uint64_t)p + ZD_MIN_LOOKAHEAD > data_end && p - base >= ZD_TILE + ZD_MAX_DIST)
                                base += ZD_TILE;
                            else
                                break;
                        }
                        /* the head of p's chain: looked at when the walk begins */
                        const uint32_t *run = job.sorted + (uint64_t)(p >> 15) * ZD_TILE;
                        LV(ws.q0) = nA ? run[(rh & 0xffffu) - 1u] : (run - ZD_TILE)[rh >> 16];
                        LV(ws.hd) = 1;
                        LV(ws.base) = base;
                        LV(ws.idx) = idx;
                        LV(ws.p) = p;
                        LV(ws.b0) = b0;
                        LV(ws.res) = MT_NONE;
                        LV(ws.nrec) = 0;
                        LV(ws.b) = b0;
                        LV(ws.bnd) = p;
                        LV(ws.where) = 0;
                        LV(ws.qs) = b0 >= 5u ? p : 0xffffffffu;
                        LV(ws.looked) = 0;
                        LV(ws.st) = MT_S_PICK;
                    }
                }
                if (!search)
                    mt_store(job, lds, kind, c, npos, idx, p, res);
            }
        }
        {
            LANEVAR(int, busy);
            FOR_LANES { LV(busy) = LV(ws.st) != MT_S_IDLE; }
            if (BALLOT(busy) == 0)
                break;
        }
        /* ---- the shortest of the chains of p, p+1, p+b-3, p+b-2 ---- */
        FOR_LANES
        {
            if (LV(ws.st) == MT_S_PICK) {
                const uint32_t p = LV(ws.p), b = LV(ws.b);
                const uint32_t mi = p - (c0 - 1u);
                uint32_t j = 0, cj = lds->mcnt[mi];
                uint32_t tj = (cj & 0xffffu) + (cj >> 16);
                if (b >= 3u) {
                    const uint32_t jm = b - 2u;
                    const uint32_t o1 = 1u, o2 = jm >= 2u ? jm - 1u : 1u, o3 = jm;
                    const uint32_t c1 = lds->mcnt[mi + o1], c2 = lds->mcnt[mi + o2], c3 = lds->mcnt[mi + o3];
                    /* (a position without three bytes ahead has no chain: staged as empty, skipped here) */
                    const uint32_t t1 = (uint64_t)p + o1 + 3u <= n ? (c1 & 0xffffu) + (c1 >> 16) : 0xffffffffu;
                    const uint32_t t2 = (uint64_t)p + o2 + 3u <= n ? (c2 & 0xffffu) + (c2 >> 16) : 0xffffffffu;
                    const uint32_t t3 = (uint64_t)p + o3 + 3u <= n ? (c3 & 0xffffu) + (c3 >> 16) : 0xffffffffu;
                    if (t1 < tj) {
                        tj = t1, cj = c1, j = o1;
                    }
                    if (t2 < tj) {
                        tj = t2, cj = c2, j = o2;
                    }
                    if (t3 < tj) {
                        tj = t3, cj = c3, j = o3;
                    }
                }
                const uint32_t rhj = lds->mrh[mi + j];
                const uint32_t ip = p - wbase;
                LV(ws.j) = j;
                LV(ws.cj) = cj;
                LV(ws.rhj) = rhj;
                LV(ws.v) = 0;
                LV(ws.sb) = (uint32_t)win[ip + b] << 8 | win[ip + b - 1u];
                const uint32_t tpos = (p + j) & ~ZD_TILE_MASK;
                LV(ws.cur) = mt_block(job, tpos, rhj & 0xffffu, rhj >> 16, cj & 0xffffu, tj, 0u);
                if (4u < tj)
                    LV(ws.nxt) = mt_block(job, tpos, rhj & 0xffffu, rhj >> 16, cj & 0xffffu, tj, 4u);
                LV(ws.st) = MT_S_WALK;
            }
        }
    }
}

#endif
