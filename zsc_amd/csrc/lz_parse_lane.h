/*
 * lz_parse_lane.h -- kernel 2 for levels 4-9, default geometry: one LANE per parse.
 *
 * lz_parse_seg.h puts 64 lanes on the candidates of ONE position and keeps the skeleton
 * of deflate_slow (reference src/deflate.c:1989-2122) in scalar registers; it is bound by
 * scalar-instruction issue (profiles/README.md: 146 scalar instructions per input byte).
 * Here every lane runs the whole of deflate_slow + longest_match (:1400-1518) for its
 * own segment of the buffer, as vector code: the skeleton costs one instruction for 64
 * parses instead of one for one.  A super-step of SL_SPAN positions is cut into SL_NS
 * segments of SL_G positions, one per lane of the workgroup, all parsed at once --
 * speculatively, exactly as in lz_parse_seg.h:
 *
 *   - chains do not depend on the parse (levels 4-9 insert every position,
 *     :2018,2069-2075) and the window base is a function of the position alone, so a
 *     parser that stands at x with no match pending (match_length == 2; whether a
 *     literal is still owed for x-1 only decides who emits that literal) goes on
 *     identically whatever came before;
 *   - every lane starts its segment in that state and records the positions of its own
 *     segment at which it is in it again (trace bit, owed-literal bit, token count);
 *   - it keeps going past the end of its segment until it is in that state at a position
 *     the owner of that position also recorded: from there on the owner's tokens ARE its
 *     tokens.  If the parse leaves the super-step first, or the lane's token area fills
 *     up, it stops with its exact state and the parse carries on from that state.
 *
 * The walk along a chain is the reference's, candidate by candidate, per lane: entry
 * from sorted[] (16 bytes = 4 entries per load), two window bytes at best_len-1 from the
 * LDS ring (:1462-1465), the first four bytes only when those match, the byte-wise
 * compare (:1485-1488) eight bytes a step only for candidates that pass.  zsc's rule that
 * a failed pre-check costs no chain budget (:1467-1468 vs :1508-1509) is just the order of
 * the tests.  Lanes are in different states at any time -- advancing the parse, walking,
 * comparing -- so the wave runs a loop of three blocks, each executed for the lanes that
 * want it; the (long) advance block is held back until enough lanes want it.
 *
 * A position whose chain is dense (a bitmap's all-zero trigram: 25 000 candidates) would
 * keep one lane busy for 25 000 steps; such searches are handed to the whole wave, which
 * sweeps the window for them (lz_parse_seg.h SG_SWEEP) -- sl_heavy().
 *
 * After the parse the hand-overs are followed from segment 0 (whose start state is the
 * true one), first the sync points in parallel (one lane per segment), then the chain
 * itself by one lane over LDS, then the token ranges are copied to the symbol stream by
 * one lane per range, which also finds where blocks are cut (every lit_bufsize-1
 * symbols, include/zsc/deflate.h:338-354).  Symbol stream and block records are
 * identical to the other parsers'.
 */
#ifndef ZSC_LZ_PARSE_LANE_H
#define ZSC_LZ_PARSE_LANE_H

#include "lz_parse_seg.h"

#ifndef SL_COUNT
#define SL_COUNT(what, n) /* event counters of the host emulation (tests/emu) */
#endif
#ifdef ZSC_WAVE_EMU
#define SL_BULK_STAT(k) (g_sl_bulk[k]++)
#else
#define SL_BULK_STAT(k)
#endif

#ifndef SL_W
#define SL_W 4 /* waves per workgroup */
#endif
#ifndef SL_G
#define SL_G 64u /* positions per segment */
#endif
#define SL_NS ((uint32_t)SL_W * 64u)   /* segments per super-step: one per lane */
#define SL_SPAN (SL_NS * SL_G)         /* positions per super-step */
#define SL_TOKCAP (SL_G + 448u)        /* tokens one parser can emit before it must stop */
#ifndef SL_ADV_MIN
#define SL_ADV_MIN 16 /* lanes that must want the advance block before it runs ... */
#endif
#ifndef SL_ADV_AGE
#define SL_ADV_AGE 12u /* ... unless some lane has waited this many rounds */
#endif
#ifndef SL_HEAVY
#define SL_HEAVY 1536u /* chains at least this long go to the whole wave (if dense enough to sweep) ... */
#endif
#ifndef SL_HEAVY_AFTER
#define SL_HEAVY_AFTER 24u /* ... once the lane has looked at this many candidates itself */
#endif
#define SL_NBULK 8u
#ifndef SL_USE_BULK
#define SL_USE_BULK 0
#endif
#ifndef SL_TAIL
#define SL_TAIL 8u /* bulk slots are for the tail of a super-step: when at most this many lanes of the wave still parse */
#endif
#ifndef SL_BULK_MIN
#define SL_BULK_MIN 16u /* a chain with at least this many candidates left (and its first block done) is walked 64 at a time */
#endif
#ifndef SL_STEPS
#define SL_STEPS 4 /* looks / compare steps a lane may take per round */
#endif
#define SL_MAXCUT (SL_SPAN / ZD_SYM_CAP + 3u)

/* lane modes */
#define SL_DONE 0u
#define SL_ADV 1u   /* at a loop top of deflate_slow */
#define SL_ADV2 2u  /* a search has ended: settle the position, then the next loop top */
#define SL_WAIT 3u  /* at a loop top past its segment, waiting for the owner of the position to get there */
#define SL_SRCH 4u  /* walking a chain */
#define SL_CMP 5u   /* comparing a candidate that passed the pre-check */
#define SL_HVY 6u   /* the search is the whole wave's */
#define SL_LOOK 7u  /* the candidate on top of the block wants the rest of the pre-check */

#define SL_EXIT_SYNCED 1u
#define SL_EXIT_STOPPED 2u /* token area full: the parse carries on from the exact state */
#define SL_EXIT_LAST 3u
#define SL_EXIT_END 4u

/* per-lane state of one wave between rounds */
typedef struct {
    LANEVAR(uint32_t, mode);
    LANEVAR(uint32_t, slot); /* its segment */
    LANEVAR(uint32_t, p);
    LANEVAR(uint32_t, cur_len);
    LANEVAR(uint32_t, cur_at);
    LANEVAR(uint32_t, pending);
    LANEVAR(uint32_t, lit);
    LANEVAR(uint32_t, ntok);
    LANEVAR(uint32_t, base);
    LANEVAR(uint32_t, data_end);
    LANEVAR(uint32_t, a_s);
    LANEVAR(uint32_t, e_s);
    LANEVAR(uint32_t, age);
    /* the position being settled */
    LANEVAR(uint32_t, prev_len);
    LANEVAR(uint32_t, prev_at);
    LANEVAR(uint32_t, look);
    LANEVAR(uint32_t, s0123);
    /* the search */
    LANEVAR(uint32_t, best);
    LANEVAR(uint32_t, where);
    LANEVAR(uint32_t, budget);
    LANEVAR(uint32_t, sb);
    LANEVAR(uint32_t, cap);
    LANEVAR(uint32_t, nice);
    LANEVAR(uint32_t, floor_pos);
    LANEVAR(uint32_t, head_seen);
    LANEVAR(uint32_t, v);
    LANEVAR(uint32_t, total);
    LANEVAR(uint32_t, nA);
    LANEVAR(int32_t, hiA);
    LANEVAR(int32_t, hiB);
    LANEVAR(uint4, E);         /* the block of entries being worked through */
    LANEVAR(uint4, N);         /* the block asked for (a load's destination: only read in step 4 of a round) */
    LANEVAR(uint32_t, left);   /* entries of E still to look at */
    LANEVAR(uint32_t, n_for);  /* the candidate number N was asked for */
    LANEVAR(uint64_t, pm1);    /* meta of position pm1_at / pmj_at, asked for ahead of the parse */
    LANEVAR(uint64_t, pmj);
    LANEVAR(uint32_t, pm1_at);
    LANEVAR(uint32_t, pmj_at);
    LANEVAR(uint32_t, hvy_ok); /* the search has not been considered for the whole wave yet */
    LANEVAR(uint32_t, bslot);  /* the bulk slot that works for this lane, or SL_NONE */
    LANEVAR(uint32_t, nbulk);  /* (the same in every lane) bulk slots in use */
    LANEVAR(uint32_t, bk0);    /* the entries of the bulk slots: lane l holds entry l of each */
    LANEVAR(uint32_t, bk1);
    LANEVAR(uint32_t, bk2);
    LANEVAR(uint32_t, bk3);
    LANEVAR(uint32_t, bk4);
    LANEVAR(uint32_t, bk5);
    LANEVAR(uint32_t, bk6);
    LANEVAR(uint32_t, bk7);
    LANEVAR(uint32_t, cq);
    LANEVAR(uint32_t, coff);
} SlWave;

typedef struct {
    uint32_t exit_p;
    uint32_t packed; /* ntok | exit kind << 16 | pending << 19 | exit_len << 20 */
} SlExit;

/* one range of tokens on the true path */
typedef struct {
    uint32_t slot_from; /* slot | first token << 16 */
    uint32_t cnt_lead;  /* tokens | (a literal for the byte before the range comes first) << 31 */
    uint32_t out_off;   /* symbols of this round before it */
    uint32_t cov;       /* input covered before it */
} SlNode;

typedef struct {
    uint32_t nsyms_after, cov_after, at;
} SlCut;

/* The candidates the lanes of one wave want looked at in a round, pooled: every searching lane
 * puts its block of entries and what the pre-check needs to know about its search here, and the
 * 64 lanes share the candidates out evenly among themselves */
typedef struct {
    uint4 blk[WAVE];            /* the lane's block: eight 16-bit entries, the next one on top */
    uint4 par[WAVE];            /* x: position of entry value 0, y: best_len - 1, z: the two bytes, w: floor */
    uint16_t own[WAVE * 8];     /* candidate i of the round belongs to lane own[i] & 255, its entry number own[i] >> 8 */
    uint32_t res[WAVE];         /* bit k: the lane's k-th candidate needs a look by its lane; bit 8 + k: it lies outside the window */
    /* long chains: the next 64 candidates of up to SL_NBULK lanes, one per lane of the wave */
    uint32_t bown[SL_NBULK];    /* the lane a slot works for, or SL_NONE */
    uint32_t bvn[SL_NBULK];     /* the candidate number of the slot's first entry */
    uint32_t bcnt[SL_NBULK];    /* entries in the slot */
    uint32_t bnew[SL_NBULK];    /* asked for in this round: looked at in the next one */
} SlStage;

struct SlLds {
    static constexpr uint32_t CHUNK = 1024u;
    static constexpr uint32_t RING = (ZD_TILE + SL_SPAN + 2u * ZD_MIN_LOOKAHEAD + 2u * CHUNK - 1u) / CHUNK * CHUNK;
    static constexpr bool HAS_INS = false;
    uint8_t ring[RING + 512];
    union {
        struct {
            uint32_t trace[SL_NS][SL_G / 32]; /* positions its own parser was at with match_length 2 */
            uint32_t tpend[SL_NS][SL_G / 32]; /* ... with a literal owed for the byte before */
        } t;
        SlNode chain[SL_NS]; /* the resolve phase's list (the traces are dead by then) */
    } u;
    SlStage stage[SL_W];    /* per wave: the candidates of one round, pooled */
    uint32_t prog[SL_NS];   /* leading positions of the segment whose trace bits are final */
    SlExit wv[SL_NS];
    uint32_t link[SL_NS]; /* next slot | first token << 10 | lead literal << 31; 0xffffffff: none */
    SlCut cuts[SL_MAXCUT];
    /* workgroup state */
    uint32_t S0;
    uint32_t finished;
    uint32_t start_p, start_len, start_at, start_pending; /* the true state at the start of the round */
    uint32_t lo, hi, wrap_base;
    uint32_t nchain, ncuts, tail_cov, round_syms;
    SgOut out;
};

typedef struct {
    uint32_t *tok;  /* SL_NS * SL_TOKCAP */
    uint16_t *sidx; /* SL_NS * SL_G: tokens emitted before a recorded position */
    uint32_t *xat;  /* SL_NS: match_start at the exit */
} SlScratch;

#define SL_SCRATCH_TOK (SL_NS * SL_TOKCAP)
#define SL_SCRATCH_SIDX (SL_NS * SL_G)
#define SL_SCRATCH_XAT (SL_NS)

DEV uint32_t sl_nact(uint32_t S0, uint32_t n)
{
    const uint32_t left = n - S0;
    if (left == 0)
        return 1;
    return left >= SL_SPAN ? SL_NS : (left + SL_G - 1) / SL_G;
}

DEV uint32_t sl_ridx(uint32_t wrap_base, uint32_t pos)
{
    uint32_t r = pos - wrap_base;
    return r >= SlLds::RING ? r - SlLds::RING : r;
}

/* the ring's bounds once it holds everything the super-step at S0 can touch */
DEV void sl_ring_target(const LzJob &job, uint32_t S0, uint32_t lo, uint32_t hi, uint32_t wb,
                        uint32_t *lo1, uint32_t *hi1, uint32_t *wb1)
{
    typedef SlLds L;
    const uint64_t want64 = (uint64_t)S0 + SL_SPAN + 2u * ZD_MIN_LOOKAHEAD;
    const uint32_t want = want64 < job.ntot ? (uint32_t)want64 : job.ntot;
    uint32_t h = hi;
    if (h < want)
        h = (want + L::CHUNK - 1u) / L::CHUNK * L::CHUNK;
    uint32_t l = lo;
    if (h - l > L::RING)
        l = h - L::RING;
    while (l - wb >= L::RING)
        wb += L::RING;
    *lo1 = l;
    *hi1 = h;
    *wb1 = wb;
}

/* phase 1a (every wave): bring the window forward, clear the traces */
DEV void sl_phase_load(const LzJob &job, SlLds *lds, int w)
{
    typedef SlLds L;
    const uint32_t S0 = UNI(lds->S0), lo = UNI(lds->lo), hi = UNI(lds->hi), wb = UNI(lds->wrap_base);
    uint32_t lo1, hi1, wb1;
    sl_ring_target(job, S0, lo, hi, wb, &lo1, &hi1, &wb1);
    /* 16 bytes per lane; positions and ring indices are multiples of 16 */
    for (uint32_t a0 = hi + (uint32_t)w * (WAVE * 16u); a0 < hi1; a0 += SL_W * WAVE * 16u) {
        FOR_LANES
        {
            const uint32_t a = a0 + (uint32_t)LANE * 16u;
            if (a < hi1) {
                uint32_t r = a - wb;
                while (r >= L::RING)
                    r -= L::RING;
                uint8_t *dst = &lds->ring[r];
                if (a + 16 <= job.ntot) {
                    COPY16(dst, job.in + a);
                } else {
                    for (uint32_t j = 0; j < 16; j++)
                        dst[j] = a + j < job.ntot ? job.in[a + j] : (uint8_t)0;
                }
            }
        }
    }
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)w * WAVE + (uint32_t)LANE; i < SL_NS * (SL_G / 32) * 2u; i += SL_W * WAVE)
            (&lds->u.t.trace[0][0])[i] = 0;
        for (uint32_t i = (uint32_t)w * WAVE + (uint32_t)LANE; i < SL_NS; i += SL_W * WAVE)
            lds->prog[i] = 0;
    }
    WAVE_SYNC();
}

/* phase 1b (wave 0, after a barrier): the new bounds, the mirror of the ring's first bytes */
DEV void sl_phase_commit(const LzJob &job, SlLds *lds, int w)
{
    typedef SlLds L;
    if (w != 0)
        return;
    const uint32_t S0 = UNI(lds->S0), lo = UNI(lds->lo), hi = UNI(lds->hi), wb = UNI(lds->wrap_base);
    uint32_t lo1, hi1, wb1;
    sl_ring_target(job, S0, lo, hi, wb, &lo1, &hi1, &wb1);
    FOR_LANES
    {
        if (LANE < 16)
            lds->ring[L::RING + LANE] = lds->ring[LANE];
    }
    ON_LANE0
    {
        lds->lo = lo1;
        lds->hi = hi1;
        lds->wrap_base = wb1;
    }
    WAVE_SYNC();
}

/* unaligned reads of the window */
#define SL_U32(pos) lds_u32(lds->ring, sl_ridx(wrap_base, (pos)))

#define SL_NONE 0xffffffffu

/* Candidate number v of the chain of the position a lane is searching for (0 = newest) is
 * entry j of its own tile's run, or of the previous tile's (negative j): hash_sort.h */
#define SL_J(v) ((v) < LV(ws.nA) ? LV(ws.hiA) - (int32_t)(v) : LV(ws.hiB) - (int32_t)((v) - LV(ws.nA)) - (int32_t)ZD_TILE)

/* A lane holds the next candidates of its chain as a block of eight 16-bit entries, the next
 * one in the top half of E.w; one candidate done = one shift */
#define SL_NEXT()                                                                             \
    do {                                                                                      \
        LV(ws.v)++;                                                                           \
        LV(ws.left)--;                                                                        \
        LV(ws.E).w = (LV(ws.E).w << 16) | (LV(ws.E).z >> 16);                                 \
        LV(ws.E).z = (LV(ws.E).z << 16) | (LV(ws.E).y >> 16);                                 \
        LV(ws.E).y = (LV(ws.E).y << 16) | (LV(ws.E).x >> 16);                                 \
        LV(ws.E).x <<= 16;                                                                    \
    } while (0)

/* One lane at a loop top of deflate_slow: settle the position a search has ended for (mode
 * SL_ADV2), then the loop top of the next position, up to the point where its search begins.
 * Written as straight-line code -- every decision a select, branches only around memory
 * operations -- because that is what the compiler turns into good vector code.
 * Nothing loaded from global memory in this call is used in this call: what a loop top needs
 * (rank, bucket bounds and chain lengths of the position: `meta`) was asked for when the
 * search before it began -- for the two positions the parse can go to from there -- and the
 * loads are issued at the very end, so that no wave ever waits for a load it has just issued. */
DEV void sl_adv_lane(const LzJob &job, SlLds *lds, const SlScratch &scr, SlWave &ws,
                     uint32_t S0, uint32_t E, uint32_t wrap_base, int _lane)
{
    (void)_lane;
    const uint32_t slot = LV(ws.slot);
    uint32_t p = LV(ws.p), cur_len = LV(ws.cur_len), cur_at = LV(ws.cur_at);
    uint32_t pending = LV(ws.pending), ntok = LV(ws.ntok), lit = LV(ws.lit);
    const int settle = LV(ws.mode) == SL_ADV2;
    uint32_t prev_len = LV(ws.prev_len), prev_at = LV(ws.prev_at), look = LV(ws.look);
    uint32_t s0123 = LV(ws.s0123);

    /* ---- what the search found (:2029-2047) ---- */
    {
        const int hs = settle && LV(ws.head_seen) != 0;
        const uint32_t fl = LV(ws.best) < look ? LV(ws.best) : look;
        cur_at = hs ? LV(ws.where) : cur_at;
        cur_len = hs ? fl : cur_len;
        const int drop = hs && cur_len <= 5u && (job.strategy == 1u || (cur_len == 3u && p - cur_at > ZD_TOO_FAR));
        cur_len = drop ? 2u : cur_len;
    }
    /* ---- the previous match, a literal, or nothing yet (:2052-2100) ---- */
    {
        const int is_match = settle && prev_len >= 3u && cur_len <= prev_len;
        const int is_lit = settle && !is_match && pending != 0;
        const uint32_t sym = is_match ? (((p - 1u - prev_at) << 16) | (prev_len - 3u)) : lit;
        if (is_match || is_lit)
            scr.tok[slot * SL_TOKCAP + ntok] = sym;
        ntok += (is_match || is_lit) ? 1u : 0u;
        pending = settle ? (is_match ? 0u : 1u) : pending;
        cur_len = is_match ? 2u : cur_len;
        p += settle ? (is_match ? prev_len - 1u : 1u) : 0u;
        lit = settle ? (s0123 & 0xffu) : lit;
    }
    /* ---- loop top at p ---- */
    uint32_t base = LV(ws.base), data_end = LV(ws.data_end);
    look = data_end - p;
    {
        /* fill_window's slide decision, :1563-1570,1589 (lz_refill) */
        const int refill = look < ZD_MIN_LOOKAHEAD;
        base = (refill && p - base >= ZD_TILE + ZD_MAX_DIST) ? base + ZD_TILE : base;
        const uint64_t end = (uint64_t)base + 2ull * ZD_TILE;
        const uint32_t de = end < job.n ? (uint32_t)end : job.n;
        data_end = refill ? de : data_end;
        look = data_end - p;
    }
    const int at_end = look == 0;
    const int neutral = cur_len == 2u;
    const int searching = look >= 3u && cur_len < job.cfg.lazy;
    const int past = p >= LV(ws.e_s);
    const int at_last = !at_end && past && p >= E;
    /* past its own segment: has the owner of p been here in the same state? */
    const int asks = !at_end && !at_last && past && neutral;
    const uint32_t t = asks ? (p - S0) / SL_G : 0u, r = asks ? (p - S0) % SL_G : 0u;
    uint32_t progv = 0, tword = 0;
    if (asks) {
        progv = LDS_LOAD_ACQ(&lds->prog[t]);
        tword = lds->u.t.trace[t][r >> 5];
    }
    const int unready = asks && progv <= r; /* the owner of p has not got there yet */
    const int synced = asks && !unready && ((tword >> (r & 31u)) & 1u) != 0;
    const int stopped = !at_end && !at_last && past && !synced && !unready && ntok + 2u >= SL_TOKCAP;
    const uint32_t exit_kind = at_end ? SL_EXIT_END : at_last ? SL_EXIT_LAST : synced ? SL_EXIT_SYNCED
                               : stopped ? SL_EXIT_STOPPED : 0u;
    /* the position's record must be here before its search can be set up */
    const int hit1 = p == LV(ws.pm1_at), hitj = p == LV(ws.pmj_at);
    const uint64_t m = hit1 ? LV(ws.pm1) : LV(ws.pmj);
    const int miss = exit_kind == 0u && !unready && searching && !hit1 && !hitj;
    const int proceed = exit_kind == 0u && !unready && !miss;

    if (exit_kind != 0u) {
        SlExit *me = &lds->wv[slot];
        me->exit_p = p;
        me->packed = ntok | (exit_kind << 16) | (pending << 19) | (cur_len << 20);
        scr.xat[slot] = cur_at;
    }
    if (exit_kind == 0u && !past) {
        /* inside its own segment: what others may hand over to (writing it twice does no harm) */
        const uint32_t rr = p - LV(ws.a_s);
        if (neutral) {
            lds->u.t.trace[slot][rr >> 5] |= 1u << (rr & 31u);
            if (pending)
                lds->u.t.tpend[slot][rr >> 5] |= 1u << (rr & 31u);
            scr.sidx[slot * SL_G + rr] = (uint16_t)ntok;
        }
        LDS_STORE_REL(&lds->prog[slot], rr + 1u);
    }
    if (exit_kind != 0u || past)
        LDS_STORE_REL(&lds->prog[slot], SL_G); /* everything in its own segment is final */

    uint32_t mode = exit_kind != 0u ? SL_DONE : unready ? SL_WAIT : miss ? SL_ADV : SL_ADV2;
    int want_block = 0;
    uint32_t want1 = SL_NONE, wantj = SL_NONE;
    if (proceed) {
        s0123 = SL_U32(p);
        prev_len = cur_len;
        prev_at = cur_at;
        cur_len = 2;
        LV(ws.head_seen) = 0;
        const uint32_t cn = (uint32_t)(m >> 32);
        const uint32_t nA = cn & 0xffffu, total = nA + (cn >> 16);
        if (searching && total != 0u) {
            LV(ws.nA) = nA;
            LV(ws.total) = total;
            LV(ws.hiA) = (int32_t)((uint32_t)m & 0xffffu) - 1;
            LV(ws.hiB) = (int32_t)(((uint32_t)m >> 16) & 0xffffu);
            LV(ws.floor_pos) = p - base > ZD_MAX_DIST ? p - ZD_MAX_DIST : base;
            LV(ws.cap) = look < 258u ? look : 258u;
            LV(ws.nice) = job.cfg.nice < look ? job.cfg.nice : look;
            LV(ws.best) = prev_len;
            LV(ws.where) = prev_at;
            LV(ws.budget) = prev_len >= job.cfg.good ? job.cfg.chain >> 2 : job.cfg.chain;
            LV(ws.v) = 0;
            LV(ws.left) = 0;
            LV(ws.hvy_ok) = 1;
            /* the two bytes the pre-check wants at best_len-1 (:1428-1429) */
            uint32_t sbw = s0123 >> (8u * (prev_len <= 3u ? prev_len - 1u : 0u));
            if (prev_len > 3u)
                sbw = SL_U32(p + prev_len - 1u);
            LV(ws.sb) = sbw & 0xffffu;
            mode = SL_SRCH;
            want_block = 1;
            /* where the parse can stand next: p+1, or behind the match that is pending */
            want1 = p + 1u;
            wantj = prev_len >= 3u ? p + prev_len - 1u : SL_NONE;
        }
    }
    LV(ws.mode) = mode;
    LV(ws.p) = p;
    LV(ws.cur_len) = cur_len;
    LV(ws.cur_at) = cur_at;
    LV(ws.pending) = pending;
    LV(ws.ntok) = ntok;
    LV(ws.lit) = lit;
    LV(ws.prev_len) = prev_len;
    LV(ws.prev_at) = prev_at;
    LV(ws.look) = look;
    LV(ws.s0123) = s0123;
    LV(ws.base) = base;
    LV(ws.data_end) = data_end;
    /* ---- the loads, for later rounds ---- */
    if (want_block) {
        const uint16_t *runA = job.sorted16 + (uint64_t)(p >> 15) * ZD_TILE;
        const int32_t j0 = SL_J(0u);
        LV(ws.N) = *(const uint4 *)(runA + (j0 & ~7));
        LV(ws.n_for) = 0;
    }
    if (miss)
        want1 = p;
    if (want1 != SL_NONE) {
        const int ok = (uint64_t)want1 + 2u < job.n;
        if (ok)
            LV(ws.pm1) = job.meta[want1];
        LV(ws.pm1_at) = ok ? want1 : SL_NONE;
    }
    if (wantj != SL_NONE) {
        const int ok = (uint64_t)wantj + 2u < job.n;
        if (ok)
            LV(ws.pmj) = job.meta[wantj];
        LV(ws.pmj_at) = ok ? wantj : SL_NONE;
    }
}

/* a candidate Q that passed the pre-check has length LEN (:1490-1512); select style */
#define SL_SETTLE(Q, LEN)                                                                     \
    do {                                                                                      \
        const uint32_t _len = (LEN) < LV(ws.cap) ? (LEN) : LV(ws.cap);                        \
        const int _improved = _len > LV(ws.best);                                             \
        LV(ws.where) = _improved ? (Q) : LV(ws.where);                                        \
        LV(ws.best) = _improved ? _len : LV(ws.best);                                         \
        const int _nice = _improved && _len >= LV(ws.nice);                                   \
        uint32_t _mode = _nice ? SL_ADV2 : SL_SRCH;                                           \
        if (!_nice) {                                                                         \
            SL_NEXT();                                                                        \
            LV(ws.budget)--;                                                                  \
            const int _over = LV(ws.budget) == 0 || LV(ws.v) == LV(ws.total);                 \
            _mode = _over ? SL_ADV2 : SL_SRCH;                                                \
            if (!_over && _improved)                                                          \
                LV(ws.sb) = SL_U32(LV(ws.p) + LV(ws.best) - 1u) & 0xffffu;                    \
        }                                                                                     \
        LV(ws.mode) = _mode;                                                                  \
    } while (0)

/* One lane: the candidate on top of its block passed the first half of the pre-check (or is
 * the chain head, which nobody has looked at yet): the rest of :1462-1512 for it */
DEV void sl_look_lane(SlLds *lds, SlWave &ws, uint32_t wrap_base, int _lane)
{
    (void)_lane;
    const uint32_t p = LV(ws.p), v = LV(ws.v);
    const uint32_t ent = LV(ws.E).w >> 16;
    const uint32_t q = (p & ~ZD_TILE_MASK) + ent - (v < LV(ws.nA) ? 0u : ZD_TILE);
    SL_COUNT(2, 1);
    const uint32_t g1 = SL_U32(q + LV(ws.best) - 1u) & 0xffffu;
    const uint32_t w0 = SL_U32(q);
    const int pass = g1 == LV(ws.sb) && ((w0 ^ LV(ws.s0123)) & 0xffffu) == 0; /* (:1462-1465) */
    if (pass) {
        /* the third byte follows from the equal hash (:1473-1480) */
        if (w0 == LV(ws.s0123) && LV(ws.cap) > 3u) {
            LV(ws.cq) = q;
            LV(ws.coff) = 4;
            LV(ws.mode) = SL_CMP;
        } else {
            SL_SETTLE(q, 3u);
        }
    } else {
        SL_NEXT();
        LV(ws.mode) = LV(ws.v) == LV(ws.total) ? SL_ADV2 : SL_SRCH;
    }
}

/* The pre-check of every candidate the wave's lanes hold in their blocks (:1462-1463: the two
 * bytes at best_len-1), shared out evenly over the 64 lanes.  Failing it is free in zsc
 * (:1467-1468), so all that matters per lane is its FIRST candidate that does not fail (or
 * that lies outside the window, :1512): everything before it is done with. */
DEV void sl_precheck(SlLds *lds, SlWave &ws, int w, uint32_t wrap_base)
{
    SlStage *sg = &lds->stage[w];
    LANEVAR(uint32_t, cnt);
    LANEVAR(uint32_t, off);
    uint32_t T = 0;
    FOR_LANES
    {
        const int owner = LV(ws.mode) == SL_SRCH && LV(ws.left) != 0u;
        LV(cnt) = owner ? LV(ws.left) : 0u;
        if (owner) {
            sg->blk[LANE] = LV(ws.E);
            uint4 pr;
            pr.x = (LV(ws.p) & ~ZD_TILE_MASK) - (LV(ws.v) < LV(ws.nA) ? 0u : ZD_TILE);
            pr.y = LV(ws.best) - 1u;
            pr.z = LV(ws.sb);
            pr.w = LV(ws.floor_pos);
            sg->par[LANE] = pr;
        }
        sg->res[LANE] = 0;
    }
    WAVE_EXSCAN(cnt, off, T);
    if (T == 0)
        return;
    FOR_LANES
    {
        for (uint32_t k = 0; k < 8u; k++) {
            if (k < LV(cnt))
                sg->own[LV(off) + k] = (uint16_t)((uint32_t)LANE | (k << 8));
        }
    }
    WAVE_SYNC();
    SL_COUNT(6, (T + 63u) / 64u);
    SL_COUNT(10, T);
    SL_COUNT(1, T);
    for (uint32_t i0 = 0; i0 < T; i0 += WAVE) {
        FOR_LANES
        {
            const uint32_t i = i0 + (uint32_t)LANE;
            if (i < T) {
                const uint32_t ow = sg->own[i];
                const uint32_t o = ow & 255u, k = ow >> 8;
                const uint32_t ent = ((const uint16_t *)&sg->blk[o])[7u - k];
                const uint4 pr = sg->par[o];
                const uint32_t q = pr.x + ent;
                const int dead = q <= pr.w; /* (such a position may have left the LDS ring: not read) */
                const uint32_t g1 = SL_U32((dead ? pr.w + 1u : q) + pr.y) & 0xffffu;
                if (dead || g1 == pr.z)
                    LDS_OR_U32(&sg->res[o], (1u << k) | (dead ? 0x100u << k : 0u));
            }
        }
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (LV(cnt) != 0u) {
            const uint32_t res = sg->res[LANE];
            const uint32_t ev = res & 0xffu;
            const uint32_t e = ev ? (uint32_t)CTZ32(ev) : LV(cnt); /* candidates before it are done with */
            /* drop e entries (0..8) from the top of the block */
            uint4 b = LV(ws.E);
            if (e & 8u)
                b.x = b.y = b.z = b.w = 0;
            if (e & 4u) {
                b.w = b.y;
                b.z = b.x;
                b.y = b.x = 0;
            }
            if (e & 2u) {
                b.w = b.z;
                b.z = b.y;
                b.y = b.x;
                b.x = 0;
            }
            if (e & 1u) {
                b.w = (b.w << 16) | (b.z >> 16);
                b.z = (b.z << 16) | (b.y >> 16);
                b.y = (b.y << 16) | (b.x >> 16);
                b.x <<= 16;
            }
            LV(ws.E) = b;
            LV(ws.v) += e;
            LV(ws.left) -= e;
            const int leaves = ev != 0u && ((res >> 8) >> e & 1u) != 0; /* the chain leaves the window (:1512) */
            LV(ws.mode) = (leaves || LV(ws.v) == LV(ws.total)) ? SL_ADV2 : (ev != 0u ? SL_LOOK : SL_SRCH);
        }
    }
}

/* One lane: eight more bytes of the compare (:1485-1488) */
DEV void sl_cmp_lane(SlLds *lds, SlWave &ws, uint32_t wrap_base, int _lane)
{
    (void)_lane;
    const uint32_t p = LV(ws.p), q = LV(ws.cq), off = LV(ws.coff);
    const uint32_t a0 = SL_U32(q + off) ^ SL_U32(p + off);
    const uint32_t a1 = SL_U32(q + off + 4u) ^ SL_U32(p + off + 4u);
    SL_COUNT(3, 1);
    const int differs = (a0 | a1) != 0;
    const uint32_t at = a0 != 0 ? off + ((uint32_t)CTZ32(a0 | 0x80000000u) >> 3)
                                : off + 4u + ((uint32_t)CTZ32(a1 | 0x80000000u) >> 3);
    const int done = differs || off + 8u >= LV(ws.cap);
    LV(ws.coff) = off + 8u;
    if (done) {
        const uint32_t len = differs ? at : LV(ws.cap);
        SL_SETTLE(q, len);
    }
}

/* The search of lane `hl` from candidate ws.cq on (not looked at yet, inside the window), by
 * the whole wave: the window sweep of lz_parse_seg.h. */
DEV void sl_heavy(const LzJob &job, SlLds *lds, SlWave &ws, uint32_t wrap_base, int hl)
{
    typedef SlLds L;
    LzState st;
    st.lo = st.hi = 0;
    st.wrap_base = wrap_base;
    st.base = READLANE(ws.base, hl);
    const uint32_t p = READLANE(ws.p, hl);
    const uint32_t s0123 = READLANE(ws.s0123, hl);
    const uint32_t floor_pos = READLANE(ws.floor_pos, hl), cap = READLANE(ws.cap, hl), nice = READLANE(ws.nice, hl);
    uint32_t best = READLANE(ws.best, hl), where = READLANE(ws.where, hl), budget = READLANE(ws.budget, hl);
    uint32_t sb = READLANE(ws.sb, hl);
    const uint32_t q0 = READLANE(ws.cq, hl);
    int fin = 0;
    LANEVAR(uint32_t, pv);
    FOR_LANES { LV(pv) = 0; }
    uint32_t pv_at = 0xffffffffu;
    SG_SWEEP(q0);
    FOR_LANES
    {
        if (LANE == hl) {
            LV(ws.best) = best;
            LV(ws.where) = where;
            LV(ws.mode) = SL_ADV2;
        }
    }
}

/* start of a parse round for wave w: lane l takes segment 64 w + l, in the fresh state at its
 * start; segment 0 starts in the true state, wherever the previous round left off */
DEV void sl_parse_start(const LzJob &job, SlLds *lds, int w, SlWave &ws)
{
    const uint32_t S0 = UNI(lds->S0);
    const uint64_t E64 = (uint64_t)S0 + SL_SPAN;
    const uint32_t E = E64 < job.n ? (uint32_t)E64 : job.n;
    const uint32_t nact = sl_nact(S0, job.n);
    const uint32_t wrap_base = UNI(lds->wrap_base);
    const uint32_t sp = UNI(lds->start_p), slen = UNI(lds->start_len), sat = UNI(lds->start_at),
                   spend = UNI(lds->start_pending);
    FOR_LANES
    {
        const uint32_t s = (uint32_t)w * WAVE + (uint32_t)LANE;
        LV(ws.mode) = SL_DONE;
        LV(ws.slot) = s;
        LV(ws.cur_len) = 2;
        LV(ws.cur_at) = 0;
        LV(ws.pending) = 0;
        LV(ws.lit) = 0;
        LV(ws.ntok) = 0;
        LV(ws.age) = 0;
        LV(ws.a_s) = S0 + s * SL_G;
        LV(ws.e_s) = LV(ws.a_s) + SL_G < E ? LV(ws.a_s) + SL_G : E;
        LV(ws.p) = LV(ws.a_s);
        LV(ws.prev_len) = LV(ws.prev_at) = LV(ws.look) = LV(ws.s0123) = 0;
        LV(ws.best) = LV(ws.where) = LV(ws.budget) = LV(ws.sb) = LV(ws.cap) = LV(ws.nice) = 0;
        LV(ws.floor_pos) = LV(ws.head_seen) = LV(ws.v) = LV(ws.total) = LV(ws.nA) = 0;
        LV(ws.hiA) = LV(ws.hiB) = 0;
        LV(ws.left) = 0;
        LV(ws.n_for) = SL_NONE;
        LV(ws.E).x = LV(ws.E).y = LV(ws.E).z = LV(ws.E).w = 0;
        LV(ws.N) = LV(ws.E);
        LV(ws.cq) = LV(ws.coff) = 0;
        LV(ws.base) = LV(ws.data_end) = 0;
        LV(ws.pm1) = LV(ws.pmj) = 0;
        LV(ws.pm1_at) = LV(ws.pmj_at) = SL_NONE;
        LV(ws.hvy_ok) = 0;
        LV(ws.bslot) = SL_NONE;
        LV(ws.nbulk) = 0;
        LV(ws.bk0) = LV(ws.bk1) = LV(ws.bk2) = LV(ws.bk3) = 0;
        LV(ws.bk4) = LV(ws.bk5) = LV(ws.bk6) = LV(ws.bk7) = 0;
        if (s == 0) {
            LV(ws.p) = sp;
            LV(ws.cur_len) = slen;
            LV(ws.cur_at) = sat;
            LV(ws.pending) = spend;
        }
        if (s < nact) {
            const uint32_t p0 = LV(ws.p);
            LV(ws.base) = sg_base_at(job.cfg, p0, job.n, ZD_MIN_LOOKAHEAD);
            const uint64_t end = (uint64_t)LV(ws.base) + 2ull * ZD_TILE;
            LV(ws.data_end) = end < job.n ? (uint32_t)end : job.n;
            if (LV(ws.pending))
                LV(ws.lit) = lds->ring[sl_ridx(wrap_base, p0 - 1u)];
            LV(ws.mode) = SL_ADV;
            if (p0 > LV(ws.a_s)) {
                /* the true state enters its segment further in: nothing before it is recorded */
                const uint32_t skip = p0 - LV(ws.a_s);
                lds->prog[s] = skip < SL_G ? skip : SL_G;
            }
            if ((uint64_t)p0 + 2u < job.n) {
                LV(ws.pm1) = job.meta[p0];
                LV(ws.pm1_at) = p0;
            }
        }
    }
    ON_LANE0
    {
        for (uint32_t i = 0; i < SL_NBULK; i++)
            lds->stage[w].bown[i] = SL_NONE;
    }
    WAVE_SYNC();
}

/* Bulk slots.  A lane whose chain is long would need a round per eight candidates; instead the
 * wave fetches the next 64 entries of such a chain with one coalesced load, lane l entry l, and
 * looks at them in the next round: one ballot finds the first candidate that does not fail the
 * pre-check (or lies outside the window), the lane takes over from there. */
#define SL_BULK_LOAD(I, BK)                                                                   \
    do {                                                                                      \
        if (UNI(sg->bnew[I]) != 0u && UNI(sg->bown[I]) != SL_NONE) {                          \
            const int _h = (int)UNI(sg->bown[I]);                                             \
            const uint32_t _vn = UNI(sg->bvn[I]), _cnt = UNI(sg->bcnt[I]);                    \
            const uint32_t _p = READLANE(ws.p, _h), _nA = READLANE(ws.nA, _h);                \
            const int32_t _hiA = READLANE(ws.hiA, _h), _hiB = READLANE(ws.hiB, _h);           \
            const uint16_t *_run = job.sorted16 + (uint64_t)(_p >> 15) * ZD_TILE;             \
            FOR_LANES                                                                         \
            {                                                                                 \
                const uint32_t _v = _vn + (uint32_t)LANE;                                     \
                const int32_t _j = _v < _nA ? _hiA - (int32_t)_v : _hiB - (int32_t)(_v - _nA) - (int32_t)ZD_TILE; \
                LV(BK) = (uint32_t)LANE < _cnt ? (uint32_t)_run[_j] : 0u;                     \
            }                                                                                 \
        }                                                                                     \
    } while (0)

#define SL_BULK_EVAL(I, BK)                                                                   \
    do {                                                                                      \
        if (UNI(sg->bown[I]) != SL_NONE && UNI(sg->bnew[I]) == 0u) {                          \
            const int _h = (int)UNI(sg->bown[I]);                                             \
            const uint32_t _vn = UNI(sg->bvn[I]), _cnt = UNI(sg->bcnt[I]);                    \
            const uint32_t _vh = READLANE(ws.v, _h), _lh = READLANE(ws.left, _h), _mh = READLANE(ws.mode, _h); \
            SL_BULK_STAT((_mh == SL_SRCH && _lh == 0u && _vh == _vn) ? 0 : _mh != SL_SRCH ? 1 : _lh != 0u ? 2 : 3); \
            if (_mh == SL_SRCH && _lh == 0u && _vh == _vn) {                                  \
                const uint32_t _p = READLANE(ws.p, _h), _nA = READLANE(ws.nA, _h);            \
                const uint32_t _qb = (_p & ~ZD_TILE_MASK) - (_vn < _nA ? 0u : ZD_TILE);       \
                const uint32_t _boff = READLANE(ws.best, _h) - 1u, _sb = READLANE(ws.sb, _h); \
                const uint32_t _fl = READLANE(ws.floor_pos, _h), _tot = READLANE(ws.total, _h); \
                LANEVAR(int, _ev);                                                            \
                LANEVAR(int, _dd);                                                            \
                SL_COUNT(6, 1);                                                               \
                SL_COUNT(10, _cnt);                                                           \
                SL_COUNT(1, _cnt);                                                            \
                FOR_LANES                                                                     \
                {                                                                             \
                    const uint32_t _q = _qb + LV(BK);                                         \
                    const int _in = (uint32_t)LANE < _cnt;                                    \
                    const int _dead = _in && _q <= _fl;                                       \
                    const uint32_t _g = SL_U32(((_dead || !_in) ? _fl + 1u : _q) + _boff) & 0xffffu; \
                    LV(_dd) = _dead;                                                          \
                    LV(_ev) = _dead || (_in && _g == _sb);                                    \
                }                                                                             \
                const uint64_t _m = BALLOT(_ev), _md = BALLOT(_dd);                           \
                const uint32_t _e = _m ? (uint32_t)CTZ64(_m) : _cnt;                          \
                const int _leaves = _m != 0 && ((_md >> _e) & 1ull) != 0;                     \
                const uint32_t _ent = READLANE(BK, _m ? _e : 0u);                             \
                const uint32_t _qlast = _qb + READLANE(BK, _cnt - 1u);                        \
                FOR_LANES                                                                     \
                {                                                                             \
                    if (LANE == _h) {                                                         \
                        LV(ws.v) = _vn + _e;                                                  \
                        if (_leaves || _vn + _e == _tot) {                                    \
                            LV(ws.mode) = SL_ADV2; /* the chain leaves the window, or ends */ \
                        } else if (_m != 0) {                                                 \
                            LV(ws.E).w = _ent << 16;                                          \
                            LV(ws.left) = 1;                                                  \
                            LV(ws.mode) = SL_LOOK;                                            \
                        } else if (LV(ws.hvy_ok) != 0u && _vn + _e >= SL_HEAVY_AFTER && _qlast > _fl + 1u && \
                                   _tot - (_vn + _e) >= SL_HEAVY &&                           \
                                   _tot - (_vn + _e) > 128u * ((_qlast - _fl) / 1024u + 2u)) { \
                            /* what is left is long and dense: the window sweep, from behind the \
                             * last candidate looked at */                                    \
                            LV(ws.hvy_ok) = 0;                                                \
                            LV(ws.cq) = _qlast - 1u;                                          \
                            LV(ws.mode) = SL_HVY;                                             \
                        }                                                                     \
                    }                                                                         \
                }                                                                             \
            }                                                                                 \
            FOR_LANES                                                                         \
            {                                                                                 \
                if (LANE == _h)                                                               \
                    LV(ws.bslot) = SL_NONE;                                                   \
            }                                                                                 \
            ON_LANE0 { sg->bown[I] = SL_NONE; }                                               \
        }                                                                                     \
    } while (0)

/* one round of wave w's loop; returns 1 when all its lanes are done */
DEV int sl_parse_round(const LzJob &job, SlLds *lds, const SlScratch &scr, int w, SlWave &ws)
{
    (void)w;
    const uint32_t S0 = UNI(lds->S0);
    const uint64_t E64 = (uint64_t)S0 + SL_SPAN;
    const uint32_t E = E64 < job.n ? (uint32_t)E64 : job.n;
    const uint32_t wrap_base = UNI(lds->wrap_base);
    LANEVAR(int, f_adv);
    LANEVAR(int, f_old);
    LANEVAR(int, f_wait);
    LANEVAR(int, f_busy);
    LANEVAR(int, f_hvy);
    FOR_LANES
    {
        const uint32_t m = LV(ws.mode);
        LV(f_adv) = m == SL_ADV || m == SL_ADV2;
        LV(f_wait) = m == SL_WAIT;
        LV(f_busy) = m == SL_SRCH || m == SL_CMP || m == SL_LOOK;
        LV(f_hvy) = m == SL_HVY;
        if (LV(f_adv))
            LV(ws.age)++;
        LV(f_old) = LV(f_adv) && LV(ws.age) >= SL_ADV_AGE;
    }
    const uint64_t m_adv = BALLOT(f_adv), m_wait = BALLOT(f_wait), m_busy = BALLOT(f_busy),
                   m_hvy = BALLOT(f_hvy), m_old = BALLOT(f_old);
    if ((m_adv | m_wait | m_busy | m_hvy) == 0)
        return 1;
    SL_COUNT(4, 1);
    if (m_hvy != 0) {
        uint64_t todo = m_hvy;
        while (todo != 0) {
            const int hl = CTZ64(todo);
            todo &= todo - 1;
            SL_COUNT(8, 1);
            sl_heavy(job, lds, ws, wrap_base, hl);
        }
    }
    /* 0: the bulk slots filled in the previous round (every load of that round has landed: its
     * last step waited for them -- nothing here waits for a load issued in this round) */
    SlStage *sg = &lds->stage[w];
#if SL_USE_BULK
    const int tail = (uint32_t)POPC64(m_adv | m_wait | m_busy | m_hvy) <= SL_TAIL;
    uint32_t nbulk = UNI(LV_UNIFORM(ws.nbulk));
    if (nbulk != 0u) {
    ON_LANE0
    {
        for (uint32_t i = 0; i < SL_NBULK; i++)
            sg->bnew[i] = 0;
    }
    WAVE_SYNC();
    SL_BULK_EVAL(0, ws.bk0);
    SL_BULK_EVAL(1, ws.bk1);
    SL_BULK_EVAL(2, ws.bk2);
    SL_BULK_EVAL(3, ws.bk3);
    SL_BULK_EVAL(4, ws.bk4);
    SL_BULK_EVAL(5, ws.bk5);
    SL_BULK_EVAL(6, ws.bk6);
    SL_BULK_EVAL(7, ws.bk7);
    WAVE_SYNC();
    nbulk = 0; /* a slot serves one round */
    }
#endif
    /* 1: the advance block, for the lanes at a loop top -- when enough of them are */
    if ((m_adv != 0 && (POPC64(m_adv) >= SL_ADV_MIN || m_old != 0 || m_busy == 0)) || (m_wait != 0 && m_busy == 0 && m_adv == 0)) {
        SL_COUNT(5, 1);
        SL_COUNT(9, (unsigned)POPC64(m_adv | m_wait));
        FOR_LANES
        {
            const uint32_t m = LV(ws.mode);
            if (m == SL_ADV || m == SL_ADV2 || m == SL_WAIT) {
                if (m == SL_WAIT)
                    LV(ws.mode) = SL_ADV;
                LV(ws.age) = 0;
                sl_adv_lane(job, lds, scr, ws, S0, E, wrap_base, LANE);
            }
        }
    }
    /* 2: ask for what a lane looks at next: 64 entries at once for a few lanes with long chains
     * (bulk slots), for the others the block of eight behind the one they are working through */
#if SL_USE_BULK
    if (tail) {
        LANEVAR(int, f_want);
        LANEVAR(int, f_long);
        FOR_LANES
        {
            const uint32_t vn = LV(ws.v) + LV(ws.left);
            LV(f_want) = LV(ws.mode) == SL_SRCH && LV(ws.bslot) == SL_NONE && LV(ws.v) != 0u &&
                         vn < LV(ws.total) && LV(ws.total) - vn >= SL_BULK_MIN;
            LV(f_long) = LV(f_want) && LV(ws.total) - vn >= 64u;
        }
        uint64_t want = BALLOT(f_want), wlong = BALLOT(f_long);
        for (uint32_t i = 0; i < SL_NBULK && want != 0; i++) {
            if (UNI(sg->bown[i]) != SL_NONE)
                continue;
            /* the longest chains gain most */
            const int h = wlong != 0 ? CTZ64(wlong) : CTZ64(want);
            want &= ~(1ull << h);
            wlong &= ~(1ull << h);
            FOR_LANES
            {
                if (LANE == h) {
                    const uint32_t vn = LV(ws.v) + LV(ws.left);
                    const uint32_t in_run = vn < LV(ws.nA) ? LV(ws.nA) - vn : LV(ws.total) - vn;
                    sg->bown[i] = (uint32_t)h;
                    sg->bvn[i] = vn;
                    sg->bcnt[i] = in_run < 64u ? in_run : 64u;
                    sg->bnew[i] = 1;
                    LV(ws.bslot) = i;
                }
            }
            WAVE_SYNC();
            nbulk++;
        }
        if (nbulk != 0u) {
        SL_BULK_LOAD(0, ws.bk0);
        SL_BULK_LOAD(1, ws.bk1);
        SL_BULK_LOAD(2, ws.bk2);
        SL_BULK_LOAD(3, ws.bk3);
        SL_BULK_LOAD(4, ws.bk4);
        SL_BULK_LOAD(5, ws.bk5);
        SL_BULK_LOAD(6, ws.bk6);
        SL_BULK_LOAD(7, ws.bk7);
        }
    }
    FOR_LANES { LV(ws.nbulk) = nbulk; }
#else
    (void)sg;
#endif
    FOR_LANES
    {
        if ((LV(ws.mode) == SL_SRCH || LV(ws.mode) == SL_CMP || LV(ws.mode) == SL_LOOK) && LV(ws.bslot) == SL_NONE) {
            const uint32_t vn = LV(ws.v) + LV(ws.left);
            if (vn < LV(ws.total) && LV(ws.n_for) != vn) {
                const uint16_t *runA = job.sorted16 + (uint64_t)(LV(ws.p) >> 15) * ZD_TILE;
                const int32_t jn = SL_J(vn);
                LV(ws.N) = *(const uint4 *)(runA + (jn & ~7));
                LV(ws.n_for) = vn;
            }
        }
    }
    /* 3: the pre-check of all the candidates in the lanes' blocks, pooled; then, lane by lane,
     * what is left to do for the candidates that passed it, and the compares */
#ifdef ZSC_WAVE_EMU
    FOR_LANES
    {
        const uint32_t m = LV(ws.mode);
        if (m != SL_DONE) {
            g_sl_lane[LV(ws.slot)]++;
            g_sl_modes[LV(ws.slot)][m == SL_ADV || m == SL_ADV2 || m == SL_WAIT ? 0 : m == SL_SRCH && LV(ws.left) == 0 ? (LV(ws.bslot) != SL_NONE ? 1 : 2) : m == SL_SRCH ? 3 : m == SL_LOOK ? 4 : m == SL_CMP ? 5 : 6]++;
        }
        SL_COUNT(12, m == SL_DONE);
        SL_COUNT(13, m == SL_ADV || m == SL_ADV2 || m == SL_WAIT);
        SL_COUNT(14, m == SL_SRCH && LV(ws.left) == 0);
        SL_COUNT(15, (m == SL_SRCH && LV(ws.left) != 0) || m == SL_CMP || m == SL_LOOK);
    }
#endif
    sl_precheck(lds, ws, w, wrap_base);
    for (int it = 0; it < SL_STEPS; it++) {
        LANEVAR(int, f_l);
        LANEVAR(int, f_c);
        FOR_LANES
        {
            LV(f_l) = LV(ws.mode) == SL_LOOK;
            LV(f_c) = LV(ws.mode) == SL_CMP;
        }
        const uint64_t m_l = BALLOT(f_l), m_c = BALLOT(f_c);
        if ((m_l | m_c) == 0)
            break;
        if (m_l != 0) {
            FOR_LANES
            {
                if (LV(ws.mode) == SL_LOOK)
                    sl_look_lane(lds, ws, wrap_base, LANE);
            }
        }
        if (m_c != 0) {
            SL_COUNT(7, 1);
            SL_COUNT(11, (unsigned)POPC64(m_c));
            FOR_LANES
            {
                if (LV(ws.mode) == SL_CMP)
                    sl_cmp_lane(lds, ws, wrap_base, LANE);
            }
        }
    }
    /* 4: lanes that have used up their block take the next one (asked for in 1 or 2); a chain's
     * first block brings its head, which has rules of its own (:2027-2028) */
    FOR_LANES
    {
        if (LV(ws.mode) == SL_SRCH && LV(ws.left) == 0 && LV(ws.n_for) == LV(ws.v)) {
            const uint32_t v = LV(ws.v);
            const int32_t j = SL_J(v);
            const uint32_t k = (uint32_t)j & 7u;
            /* the entry at j comes to the top: drop the 7 - k entries above it */
            uint4 e = LV(ws.N);
            const uint32_t sh = 7u - k;
            if (sh & 4u) {
                e.w = e.y;
                e.z = e.x;
                e.y = e.x = 0;
            }
            if (sh & 2u) {
                e.w = e.z;
                e.z = e.y;
                e.y = e.x;
                e.x = 0;
            }
            if (sh & 1u) {
                e.w = (e.w << 16) | (e.z >> 16);
                e.z = (e.z << 16) | (e.y >> 16);
                e.y = (e.y << 16) | (e.x >> 16);
                e.x <<= 16;
            }
            LV(ws.E) = e;
            /* entries of this block that belong to the chain: down to the start of the block,
             * or to the end of the run in this tile */
            const uint32_t in_run = v < LV(ws.nA) ? LV(ws.nA) - v : LV(ws.total) - v;
            LV(ws.left) = k + 1u < in_run ? k + 1u : in_run;
            LV(ws.n_for) = SL_NONE;
            const uint32_t p = LV(ws.p);
            const uint32_t q = (p & ~ZD_TILE_MASK) + (e.w >> 16) - (v < LV(ws.nA) ? 0u : ZD_TILE);
            if (v == 0u) {
                /* the chain head may lie at exactly MAX_DIST (:2027-2028), later links may not (:1512) */
                const int head_ok = q > LV(ws.base) && p - q <= ZD_MAX_DIST; /* else longest_match is not called */
                LV(ws.head_seen) = head_ok ? 1u : 0u;
                LV(ws.mode) = (!head_ok || LV(ws.best) >= LV(ws.look)) ? SL_ADV2 : SL_LOOK;
            } else if (v >= SL_HEAVY_AFTER && LV(ws.hvy_ok) != 0u && q > LV(ws.floor_pos)) {
                /* still searching: if what is left of the chain is long and dense, the rest goes to
                 * the whole wave, which sweeps the window from this candidate on instead of walking */
                LV(ws.hvy_ok) = 0;
                if (LV(ws.total) - v >= SL_HEAVY &&
                    LV(ws.total) - v > 128u * ((q - LV(ws.floor_pos)) / 1024u + 2u)) {
                    LV(ws.cq) = q;
                    LV(ws.mode) = SL_HVY;
                }
            }
        }
    }
    return 0;
}

/* ---- the resolve phase ------------------------------------------------------------- */

/* R1 (every lane for its own slot): where does its parse hand over, and at which token? */
DEV void sl_resolve_links(const LzJob &job, SlLds *lds, const SlScratch &scr, int w)
{
    const uint32_t S0 = UNI(lds->S0);
    const uint32_t nact = sl_nact(S0, job.n);
    FOR_LANES
    {
        for (uint32_t s = (uint32_t)w * WAVE + (uint32_t)LANE; s < nact; s += SL_W * WAVE) {
            uint32_t link = 0xffffffffu;
            const uint32_t pk = lds->wv[s].packed;
            if (((pk >> 16) & 7u) == SL_EXIT_SYNCED) {
                const uint32_t xp = lds->wv[s].exit_p;
                const uint32_t t = (xp - S0) / SL_G, r = (xp - S0) % SL_G;
                const uint32_t ps = (lds->u.t.tpend[t][r >> 5] >> (r & 31u)) & 1u;
                const uint32_t pp = (pk >> 19) & 1u;
                /* both owe the literal for xp-1: the owner's stream has it next; only the
                 * owner does: the byte is covered already, skip it; only this parse does: it
                 * comes first */
                const uint32_t ft = (uint32_t)scr.sidx[t * SL_G + r] + ((!pp && ps) ? 1u : 0u);
                link = t | (ft << 10) | ((pp && !ps) ? 0x80000000u : 0u);
            }
            lds->link[s] = link;
        }
    }
    WAVE_SYNC();
}

/* R2 (wave 0, after a barrier; the traces are dead from here on): follow the hand-overs */
DEV void sl_resolve_chain(const LzJob &job, SlLds *lds, int w)
{
    (void)job;
    if (w != 0)
        return;
    ON_LANE0
    {
        uint32_t k = 0, ft = 0, lead = 0;
        uint32_t out_off = 0, n = 0;
        /* input covered before the first range: what the stream covers so far */
        uint32_t cov = lds->out.cov;
        for (;;) {
            const uint32_t pk = lds->wv[k].packed;
            const uint32_t ntok = pk & 0xffffu;
            const uint32_t cnt = ntok - ft;
            SlNode *nd = &lds->u.chain[n++];
            nd->slot_from = k | (ft << 16);
            nd->cnt_lead = cnt | (lead << 31);
            nd->out_off = out_off;
            nd->cov = cov;
            out_off += cnt + lead;
            const uint32_t link = lds->link[k];
            if (link == 0xffffffffu || n >= SL_NS)
                break;
            /* the next range starts where this parse handed over: everything before that
             * position is covered, except the byte a literal is still owed for */
            cov = lds->wv[k].exit_p - ((pk >> 19) & 1u);
            k = link & 0x3ffu;
            ft = (link >> 10) & 0x1fffffu;
            lead = link >> 31;
        }
        lds->nchain = n;
        lds->round_syms = out_off;
        lds->ncuts = 0;
    }
    WAVE_SYNC();
}

/* R3 (one lane per range, after a barrier): copy the tokens, find the block cuts */
DEV void sl_resolve_copy(const LzJob &job, SlLds *lds, const SlScratch &scr, int w)
{
    const uint32_t nchain = UNI(lds->nchain);
    const uint32_t nsyms0 = UNI(lds->out.nsyms), blk_sym0 = UNI(lds->out.blk_sym0);
    const uint32_t wrap_base = UNI(lds->wrap_base);
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)w * WAVE + (uint32_t)LANE; i < nchain; i += SL_W * WAVE) {
            const SlNode nd = lds->u.chain[i];
            const uint32_t slot = nd.slot_from & 0xffffu, from = nd.slot_from >> 16;
            uint32_t cnt = nd.cnt_lead & 0x7fffffffu;
            const uint32_t lead = nd.cnt_lead >> 31;
            const uint32_t *src = scr.tok + slot * SL_TOKCAP + from;
            uint32_t g = nsyms0 + nd.out_off; /* index of the next symbol in the buffer's stream */
            uint32_t cov = nd.cov;
            /* symbols until the block is full (include/zsc/deflate.h:338-354) */
            uint32_t room = ZD_SYM_CAP - (g - blk_sym0) % ZD_SYM_CAP;
            for (uint32_t j = 0; j < cnt + lead; j++) {
                uint32_t t;
                if (lead && j == 0)
                    t = lds->ring[sl_ridx(wrap_base, cov)]; /* the literal this parse still owed */
                else
                    t = src[j - lead];
                job.syms[g++] = t;
                const uint32_t start = cov;
                cov += (t >> 16) ? (t & 0xffu) + 3u : 1u;
                if (--room == 0) {
                    /* the iteration that emitted the token ran one position later: the block
                     * is cut there (lz_parse_seg.h sg_append) */
                    const uint32_t c = ((g - blk_sym0) / ZD_SYM_CAP) - 1u - ((nsyms0 - blk_sym0) / ZD_SYM_CAP);
                    if (c < SL_MAXCUT) {
                        lds->cuts[c].nsyms_after = g;
                        lds->cuts[c].cov_after = cov;
                        lds->cuts[c].at = start + 1u;
                    }
                    LDS_ADD_U32(&lds->ncuts, 1u);
                    room = ZD_SYM_CAP;
                }
            }
            if (i + 1u == nchain)
                lds->tail_cov = cov;
        }
    }
    WAVE_SYNC();
}

/* R4 (wave 0, after a barrier): block records, the end of the round */
DEV void sl_resolve_finish(const LzJob &job, SlLds *lds, const SlScratch &scr, int w)
{
    if (w != 0)
        return;
    ON_LANE0
    {
        SgOut *o = &lds->out;
        const uint32_t ncuts = lds->ncuts < SL_MAXCUT ? lds->ncuts : SL_MAXCUT;
        for (uint32_t c = 0; c < ncuts; c++) {
            const SlCut cu = lds->cuts[c];
            ZdBlockRec *b = &job.blocks[o->nblocks];
            const uint32_t base = sg_base_at(job.cfg, cu.at, job.n, ZD_MIN_LOOKAHEAD);
            const uint64_t wend = (uint64_t)base + 2ull * ZD_TILE;
            b->sym_begin = o->blk_sym0;
            b->sym_count = ZD_SYM_CAP;
            b->in_begin = o->blk_in0;
            b->in_len = cu.cov_after - o->blk_in0;
            b->stored_ok = o->blk_in0 >= base ? 1u : 0u;
            b->last = 0;
            b->cut = ZD_CUT_FULL;
            b->wend = wend < 0xffffffffull ? (uint32_t)wend : 0xffffffffu;
            b->at = cu.at;
            o->nblocks++;
            o->blk_sym0 = cu.nsyms_after;
            o->blk_in0 = cu.cov_after;
        }
        o->nsyms += lds->round_syms;
        o->cov = lds->tail_cov;
        /* how the last parse on the path ended */
        const SlNode last = lds->u.chain[lds->nchain - 1u];
        const uint32_t k = last.slot_from & 0xffffu;
        const uint32_t pk = lds->wv[k].packed, xp = lds->wv[k].exit_p;
        const uint32_t kind = (pk >> 16) & 7u, xpend = (pk >> 19) & 1u, xlen = pk >> 20;
        if (kind == SL_EXIT_LAST || kind == SL_EXIT_STOPPED) {
            lds->start_p = xp;
            lds->start_len = xlen;
            lds->start_at = scr.xat[k];
            lds->start_pending = xpend;
            /* a parse that had to stop inside the super-step (token area full) carries on as
             * segment 0 of a super-step that starts at its segment */
            lds->S0 += kind == SL_EXIT_LAST ? SL_SPAN : (xp - lds->S0) / SL_G * SL_G;
        } else {
            /* SL_EXIT_END: the end of the input (src/deflate.c:2108-2117) */
            if (xpend) {
                /* the last byte goes out as a literal; _tr_tally's "block full" answer is
                 * ignored here (:2109-2112), so no cut */
                job.syms[o->nsyms++] = job.in[xp - 1u];
                o->cov++;
            }
            ZdBlockRec *b = &job.blocks[o->nblocks];
            b->sym_begin = o->blk_sym0;
            b->sym_count = o->nsyms - o->blk_sym0;
            b->in_begin = o->blk_in0;
            b->in_len = job.n - o->blk_in0;
            b->stored_ok = o->blk_in0 >= sg_base(job.cfg, job.n, job.n) ? 1u : 0u;
            b->last = 1u;
            b->cut = ZD_CUT_END;
            b->wend = 0xffffffffu;
            b->at = job.n;
            o->nblocks++;
            o->blk_sym0 = o->nsyms;
            o->blk_in0 = job.n;
            job.out->nsyms = o->nsyms;
            job.out->nblocks = o->nblocks;
            lds->finished = 1;
        }
    }
    WAVE_SYNC();
}

/* before the first super-step (wave 0) */
DEV void sl_init(SlLds *lds, int w)
{
    if (w != 0)
        return;
    ON_LANE0
    {
        lds->S0 = 0;
        lds->finished = 0;
        lds->start_p = 0;
        lds->start_len = 2;
        lds->start_at = 0;
        lds->start_pending = 0;
        lds->lo = lds->hi = lds->wrap_base = 0;
        lds->nchain = lds->ncuts = lds->tail_cov = lds->round_syms = 0;
        lds->out.nsyms = lds->out.nblocks = lds->out.blk_sym0 = lds->out.blk_in0 = lds->out.cov = 0;
    }
    WAVE_SYNC();
}

#endif
