/*
 * lz_parse_seg.h -- kernel 2 for long buffers: several wavefronts parse ONE buffer.
 *
 * The lazy parse (reference deflate_slow, src/deflate.c:1989-2122) is serial, and its
 * 32 KiB window has to sit in LDS, so a wave-per-buffer kernel gets four waves onto a
 * CU and is bound by single-wave latency.  This kernel shares one window between the
 * SG_W waves of a workgroup.  The buffer advances in super-steps of SG_SPAN positions,
 * cut into segments of SG_G positions that the waves take from a work queue in LDS, last
 * segment first, and parse at the same time -- speculatively, because the state the
 * serial parse would arrive in at the start of a segment is not known yet:
 *
 *   - hash chains do not depend on the parse at levels 4-9 (every position is
 *     inserted, src/deflate.c:2018,2069-2075), and the window base is a function
 *     of the position alone, so a parser that stands at position x with NO match
 *     pending ("fresh": match_available == 0, match_length == 2) continues
 *     identically whatever happened before x;
 *     and so does one with no match pending that still owes the literal of the byte before
 *     x ("neutral": match_available == 1, match_length == 2 -- the state of every position
 *     inside a run of literals; without it incompressible data never hands over);
 *   - every segment's parser starts fresh at the segment start and records every
 *     position it visits fresh or neutral (two bitmaps in LDS, plus its token count there);
 *   - a parser keeps going past the end of its segment until it stands at a position in
 *     the state the parser of the segment it has run into also had there: from there on that
 *     parser's tokens ARE the serial parse's tokens.  Text resyncs within a few tokens;
 *   - if no common position turns up within SG_OV bytes (a long run, say), the parser
 *     stops with its exact state, and once the chain of hand-overs has reached it wave
 *     0 parses on from that state (a redo round) -- the super-step degrades towards
 *     the serial parse, never to a different result.
 *
 * After the parse phase wave 0 walks the chain of hand-overs from segment 0 (whose
 * start state is the true one, carried over from the previous super-step) and
 * appends the valid token ranges to the buffer's symbol stream, cutting blocks
 * every lit_bufsize-1 symbols exactly as _tr_tally does (include/zsc/deflate.h:338-354).
 * The symbol stream, block records and therefore every later kernel and the final
 * bytes are identical to the wave-per-buffer parser's.  The search itself is described
 * further down (SG_EVAL, SG_SWEEP).
 */
#ifndef ZSC_LZ_PARSE_SEG_H
#define ZSC_LZ_PARSE_SEG_H

#include "lz_parse.h"

/* -DSG_GROUP=16 builds this parser for four 16-lane groups per wavefront (wave_group.h): four
 * segments parsed at once by one wave, sharing every instruction they execute at the same time */
#ifdef SG_GROUP
#undef ZSC_GROUP
#define ZSC_GROUP SG_GROUP
#include "wave_group.h"
#endif

#ifndef SG_COUNT
#define SG_COUNT(what, n) /* event counters of the host emulation (tests/emu) */
#endif

#ifndef SG_W
#define SG_W 8                      /* waves per workgroup */
#endif
#ifndef SG_G
#define SG_G 256u                   /* positions per segment */
#endif
#ifndef SG_SPAN
#define SG_SPAN 8192u               /* positions per super-step */
#endif
#ifndef SG_MIN_WAVES
#define SG_MIN_WAVES 6 /* waves per SIMD the register allocation aims at (three workgroups per CU) */
#endif
#define SG_NS (SG_SPAN / SG_G)      /* segments per super-step, handed to the waves by a work queue */
#ifndef SG_OV
#define SG_OV 512u                  /* how far past its segment a parser looks for a hand-over */
#endif
#define SG_TRACE SG_G               /* positions a segment records (its own) */
#define SG_TOKCAP (SG_G + SG_OV + 320u) /* tokens one segment's parser can emit */
/* scratch of one workgroup in 32-bit words: SG_NS token areas, then SG_NS x SG_TRACE 16-bit token indices */
#define SG_SCRATCH_WORDS (SG_NS * SG_TOKCAP + SG_NS * SG_TRACE / 2u)

#ifndef SG_PICK_AHEAD
#define SG_PICK_AHEAD 16u /* positions from p on whose chain lengths a search must find in the register cache */
#endif
#ifndef SG_EMPTY_SKIP
#define SG_EMPTY_SKIP 0 /* (measured: text and incompressible data lose 3-5 % to the test, the table class gains 4 %) a pending match and an empty chain among the trigrams a longer one must contain: no search */
#endif
#ifndef SG_FAR_OVER
#define SG_FAR_OVER 128u /* the shortest chain in reach is longer than this: look at the far ones too */
#endif
#ifndef SG_STAIR_MIN
#define SG_STAIR_MIN 256u /* chains at least this long are searched as a staircase (LzJob.stair_min) */
#endif
#ifndef SG_ONE
#define SG_ONE 1 /* 0: chains below the chain budget are walked like the others */
#endif
#ifdef ZSC_WAVE_EMU
extern int g_sg_one; /* the host emulation can switch the lane-parallel search off (tests of the walk) */
#define SG_ONE_ON g_sg_one
#else
#define SG_ONE_ON 1
#endif
#ifndef SG_STAIR
#define SG_STAIR 1 /* 0: every search is the reference's walk along p's own chain (the parser as it was) */
#endif

#define SG_EXIT_SYNCED 1u
#define SG_EXIT_UNSYNCED 2u
#define SG_EXIT_LAST 3u
#define SG_EXIT_END 4u

/* the symbol stream of one buffer while it is being assembled */
typedef struct {
    uint32_t nsyms, nblocks, blk_sym0, blk_in0, cov;
} SgOut;

typedef struct {
    uint32_t start_p, start_len, start_at, start_pending;
    uint32_t exit_kind, exit_p, exit_len, exit_at, exit_pending;
    uint32_t ntok;                                  /* tokens its parser emitted */
} SgWave;

struct SgLds {
    /* window (32 KiB back) + one super-step + overlap + two lookaheads + one chunk being
     * loaded, in whole chunks: 45056 for the default super-step */
    static constexpr uint32_t CHUNK = 2048u;
    static constexpr uint32_t RING = (ZD_TILE + SG_SPAN + SG_OV + 2u * ZD_MIN_LOOKAHEAD + 2u * CHUNK - 1u) / CHUNK * CHUNK;
    static constexpr bool HAS_INS = false;
    static constexpr bool GLOBAL_WIN = false;
    uint8_t ring[RING + 512];
    uint32_t trace[SG_NS][SG_TRACE / 32];
    uint32_t tkind[SG_NS][SG_TRACE / 32]; /* for the positions in trace: 1 = a literal was owed there */
    SgWave wv[SG_NS];             /* one record per segment of the super-step */
    /* workgroup state */
    uint32_t S0;                  /* first position of the current super-step */
    uint32_t finished;
    uint32_t queue;               /* segments not handed out yet (taken from the top) */
    uint32_t redo, redo_seg, redo_from; /* a segment to parse again from segment redo_from's exit state */
    uint32_t chain, chain_ft;     /* where the resolver stands: segment and first valid token */
    uint32_t emu_ascending;       /* test hook of the host emulation: hand segments out bottom-up */
    uint32_t lo, hi, wrap_base;   /* window ring */
    SgOut out;
    uint32_t cstage[WAVE];
};

/* scratch in HBM per workgroup */
typedef struct {
    uint32_t *tok;  /* SG_NS * SG_TOKCAP tokens */
    uint16_t *sidx; /* SG_NS * SG_TRACE: tokens emitted before a fresh position */
} SgScratch;

/* the window base the serial parse has at a loop top at position p: every slide of
 * fill_window (src/deflate.c:1563-1570) whose condition holds at p has happened */
/* `need`: the parser calls fill_window when fewer than `need` bytes of lookahead are left --
 * MIN_LOOKAHEAD for deflate_slow/_fast (:1898,2002), MAX_MATCH+1 for deflate_rle (:2141),
 * 1 for deflate_huff (:2218) */
DEV uint32_t sg_base_at(const ZdLevel &cfg, uint32_t p, uint32_t n, uint32_t need)
{
    uint32_t base = 0;
    for (;;) {
        uint64_t end = (uint64_t)base + 2ull * cfg.wsize;
        uint32_t data_end = end < n ? (uint32_t)end : n;
        if ((uint64_t)p + need > data_end && p - base >= cfg.wsize + cfg.max_dist)
            base += cfg.wsize;
        else
            return base;
    }
}
DEV uint32_t sg_base(const ZdLevel &cfg, uint32_t p, uint32_t n)
{
    return sg_base_at(cfg, p, n, ZD_MIN_LOOKAHEAD);
}

/* What k_link_prev (hash_sort.h: hs_link_prev) would have written for position x, whose rank is rk:
 * rank | hib << 16 and cnt, from the bucket directories of x's tile and of the one before it.  The
 * parser asks for 64 consecutive positions at a time, so this is two gathers per 48 positions in
 * place of a kernel that does the same gathers for every position and writes the answers out. */
template <class L>
DEV void sg_link(const LzJob &job, const L *lds, const LzState &st, uint32_t x, uint32_t rk, uint32_t &rh,
                 uint32_t &cn)
{
    const uint32_t w = lds_u32(lds->ring, lz_ridx<L>(st, x));
    const uint32_t h = (((w & 0xffu) << 10) ^ (((w >> 8) & 0xffu) << 5) ^ ((w >> 16) & 0xffu)) & ZD_HASH_MASK;
    const uint32_t t = x >> 15;
    const uint16_t *d = job.dir + (uint64_t)t * ZD_DIR_STRIDE;
    uint32_t c = rk - (uint32_t)d[h], hb = 0xffffu;
    if (t != 0u) {
        const uint32_t two = ld_u32((const uint8_t *)(d - ZD_DIR_STRIDE + h)); /* dir_prev[h], dir_prev[h + 1] */
        hb = ((two >> 16) - 1u) & 0xffffu; /* 0xffff: bucket empty from the start */
        c |= ((two >> 16) - (two & 0xffffu)) << 16;
    }
    rh = rk | (hb << 16);
    cn = c;
}

/* number of segments of the super-step that starts at S0 */
DEV uint32_t sg_nact(uint32_t S0, uint32_t n)
{
    const uint32_t left = n - S0;
    if (left == 0)
        return 1; /* the empty buffer: segment 0 still reports the end of the input */
    return left >= SG_SPAN ? SG_NS : (left + SG_G - 1) / SG_G;
}

/* phase 1 (wave 0): slide the window to the new super-step, fill the work queue */
DEV void sg_phase_begin(const LzJob &job, SgLds *lds, int w)
{
    if (w != 0)
        return;
    typedef SgLds L;
    LzState st;
    st.lo = UNI(lds->lo);
    st.hi = UNI(lds->hi);
    st.wrap_base = UNI(lds->wrap_base);
    const uint32_t S0 = UNI(lds->S0);
    uint64_t want64 = (uint64_t)S0 + SG_SPAN + SG_OV + 2u * ZD_MIN_LOOKAHEAD;
    const uint32_t want = want64 < job.ntot ? (uint32_t)want64 : job.ntot;
    while (st.hi < want)
        lz_load_chunk<L>(job, lds, st);
    ON_LANE0
    {
        lds->lo = st.lo;
        lds->hi = st.hi;
        lds->wrap_base = st.wrap_base;
        lds->queue = sg_nact(S0, job.n);
        lds->redo = 0;
        lds->chain = 0;
        lds->chain_ft = 0;
    }
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)LANE; i < SG_NS * (SG_TRACE / 32); i += WAVE)
            (&lds->trace[0][0])[i] = 0;
    }
    WAVE_SYNC();
}

/* ---- the search ------------------------------------------------------------------
 *
 * longest_match (src/deflate.c:1414-1521) for one position, by the whole wave.  The
 * chain of p is the run of its hash bucket below rank[p] in its own tile's sorted
 * array, continued by the bucket's run in the previous tile; cnt[p] holds both run
 * lengths, so candidate number v (0 = newest) has a known address and 64 of them are
 * one coalesced load.  Per candidate the wave does what the reference's pre-check does
 * (:1462-1469): two window reads and two compares.  The four bytes read at the
 * candidate also tell whether it can be longer than three bytes at all; only those
 * that can get the cooperative 256-byte compare.  Candidates that pass the pre-check
 * but cannot improve on best_len only cost chain budget, which is settled for all of
 * them at once with a popcount. */

/* four bytes of the window as a wave-uniform value */
#define SG_PEEK32(pos, out) ((out) = GUNI(lds_u32(lds->ring, lz_ridx<L>(st, (pos)))))
/* ... at p + OFF (OFF below LZ_MIRROR): rp is p's place in the ring, and what follows it there is linear */
#define SG_PEEKP(OFF, out) ((out) = GUNI(lds_u32(lds->ring, rp + (OFF))))

/* entries of candidates 64*B .. 64*B+63; lanes past the end of the chain read entry 0 of
 * the tile (a valid address) and are masked later -- cheaper than predicating the load */
#define SG_LOAD(E, B)                                                                         \
    FOR_GLANES                                                                                 \
    {                                                                                         \
        const uint32_t _v = (B)*GRP + (uint32_t)GLANE;                                         \
        int32_t _i = _v < nA ? hiA - (int32_t)_v : hiB - (int32_t)(_v - nA) - (int32_t)ZD_TILE; \
        _i = _v < total ? _i : 0;                                                             \
        LV(E) = job.sorted[(uint32_t)((int32_t)tileA + _i)]; /* (the tile's run starts at sorted + tileA) */ \
    }

/* the string at p, four bytes a lane (dword l = bytes 4l .. 4l+3) */
#define SG_PV_LOAD()                                                                          \
    do {                                                                                      \
        if (pv_at != p) {                                                                     \
            pv_at = p;                                                                        \
            FOR_GLANES { LV(pv) = lds_u32(lds->ring, rp + 4u * (uint32_t)GLANE); }             \
        }                                                                                     \
    } while (0)

/* cooperative longest common prefix of the strings at QJ and p (at most cap bytes) */
#define SG_LCP(QJ, LEN)                                                                       \
    do {                                                                                      \
        SG_COUNT(1, 1);                                                                       \
        SG_PV_LOAD();                                                                         \
        const uint32_t _rq = lz_ridx<L>(st, (QJ)); /* (the ring is linear from there: LZ_MIRROR) */ \
        if (4u * GRP >= 256u) {                                                               \
            /* a 64-lane group: the first 256 bytes in one step, against the string at p held \
             * in registers, then the last two bytes one by one */                            \
            LANEVAR(uint32_t, _diff);                                                         \
            LANEVAR(int, _differs);                                                           \
            FOR_GLANES                                                                        \
            {                                                                                 \
                LV(_diff) = lds_u32(lds->ring, _rq + 4u * (uint32_t)GLANE) ^ LV(pv);          \
                LV(_differs) = LV(_diff) != 0;                                                \
            }                                                                                 \
            const uint64_t _dm = GBALLOT(_differs);                                           \
            if (_dm != 0) {                                                                   \
                const int _f = CTZ64(_dm);                                                    \
                (LEN) = 4u * (uint32_t)_f + ((uint32_t)CTZ32(GREADLANE(_diff, _f)) >> 3);     \
            } else {                                                                          \
                (LEN) = 256;                                                                  \
                /* (rare: nothing of this is worked out ahead of the loops around it) */      \
                uint32_t _rp = rp, _cap = cap;                                                \
                GOPAQUE(_rp);                                                                 \
                GOPAQUE(_cap);                                                                \
                if (_cap > 256 && GUNI(lds->ring[_rq + 256]) == GUNI(lds->ring[_rp + 256])) { \
                    (LEN) = 257;                                                              \
                    if (_cap > 257 && GUNI(lds->ring[_rq + 257]) == GUNI(lds->ring[_rp + 257])) \
                        (LEN) = 258;                                                          \
                }                                                                             \
            }                                                                                 \
        } else {                                                                              \
            /* a narrower group: 4 GRP bytes per step; what is read past cap is cut off below */ \
            (LEN) = cap;                                                                      \
            for (uint32_t _o = 0; _o < cap; _o += 4u * GRP) {                                 \
                LANEVAR(uint32_t, _diff);                                                     \
                LANEVAR(int, _differs);                                                       \
                FOR_GLANES                                                                    \
                {                                                                             \
                    const uint32_t _pw = _o == 0u ? LV(pv)                                    \
                                                  : lds_u32(lds->ring, rp + _o + 4u * (uint32_t)GLANE); \
                    LV(_diff) = lds_u32(lds->ring, _rq + _o + 4u * (uint32_t)GLANE) ^ _pw;    \
                    LV(_differs) = LV(_diff) != 0;                                            \
                }                                                                             \
                const uint64_t _dm = GBALLOT(_differs);                                       \
                if (_dm != 0) {                                                               \
                    const int _f = CTZ64(_dm);                                                \
                    (LEN) = _o + 4u * (uint32_t)_f + ((uint32_t)CTZ32(GREADLANE(_diff, _f)) >> 3); \
                    break;                                                                    \
                }                                                                             \
            }                                                                                 \
        }                                                                                     \
        if ((LEN) > cap)                                                                      \
            (LEN) = cap;                                                                      \
    } while (0)

/* LZ_HEAD_BLOCKED (lz_parse.h) for a group: is there a position between Q and P whose hash under
 * the reference's hash function (another mem_level) equals P's? */
#define SG_HEAD_BLOCKED(Q, P, MEMB, blocked)                                                  \
    do {                                                                                      \
        (blocked) = 0;                                                                        \
        const uint32_t _hs = (job.cfg.hbits + 2u) / 3u, _hm = (1u << job.cfg.hbits) - 1u;     \
        const uint32_t _wp = GUNI(lds_u32(lds->ring, lz_ridx<L>(st, (P))));                   \
        const uint32_t _hp = (((_wp & 0xffu) << (2u * _hs)) ^ (((_wp >> 8) & 0xffu) << _hs) ^ \
                              ((_wp >> 16) & 0xffu)) & _hm;                                   \
        for (uint32_t _x0 = (Q) + 1u; _x0 < (P) && !(blocked); _x0 += GRP) {                  \
            LANEVAR(int, _same);                                                              \
            FOR_GLANES                                                                        \
            {                                                                                 \
                const uint32_t _x = _x0 + (uint32_t)GLANE;                                    \
                int _s = 0;                                                                   \
                if (_x < (P)) {                                                               \
                    const uint32_t _w = lds_u32(lds->ring, lz_ridx<L>(st, _x));               \
                    const uint32_t _h = (((_w & 0xffu) << (2u * _hs)) ^                       \
                                         (((_w >> 8) & 0xffu) << _hs) ^ ((_w >> 16) & 0xffu)) & _hm; \
                    _s = _h == _hp && MEMB(_x);                                               \
                }                                                                             \
                LV(_same) = _s;                                                               \
            }                                                                                 \
            if (GBALLOT(_same) != 0)                                                          \
                (blocked) = 1;                                                                \
        }                                                                                     \
    } while (0)

/* evaluate candidates 64*B .. 64*B+63 (entries E); sets fin when the search is over.
 * The chain head (candidate 0) has been checked by the caller. */
#define SG_EVAL(E, B)                                                                         \
    do {                                                                                      \
        LANEVAR(uint32_t, _q);                                                                \
        LANEVAR(uint32_t, _w0);                                                               \
        LANEVAR(int, _alive);                                                                 \
        LANEVAR(int, _dead);                                                                  \
        LANEVAR(int, _pass);                                                                  \
        LANEVAR(int, _maybe);                                                                 \
        FOR_GLANES                                                                             \
        {                                                                                     \
            const uint32_t _v = (B)*GRP + (uint32_t)GLANE;                                     \
            const uint32_t q = tileA + (LV(E) & ZD_TILE_MASK) - (_v < nA ? 0u : ZD_TILE);     \
            LV(_q) = q;                                                                       \
            LV(_alive) = _v == 0 || (_v < total && q > floor_pos); /* :1519 */                \
            LV(_dead) = _v != 0 && _v < total && q <= floor_pos;                              \
        }                                                                                     \
        if (job.cfg.hbits != 15u) {                                                           \
            /* with another mem_level the candidate at exactly MAX_DIST can be the head of the \
             * reference's chain without being the first one here (LZ_HEAD_BLOCKED) */        \
            LANEVAR(int, _edge);                                                              \
            FOR_GLANES                                                                         \
            {                                                                                 \
                LV(_edge) = LV(_dead) && LV(_q) > st.base && p - LV(_q) == job.cfg.max_dist;  \
            }                                                                                 \
            if (GBALLOT(_edge) != 0) {                                                         \
                int _blk;                                                                     \
                SG_HEAD_BLOCKED(p - job.cfg.max_dist, p, LZ_MEMB_ALL, _blk);                  \
                if (!_blk) {                                                                  \
                    FOR_GLANES                                                                 \
                    {                                                                         \
                        if (LV(_edge)) {                                                      \
                            LV(_alive) = 1;                                                   \
                            LV(_dead) = 0;                                                    \
                        }                                                                     \
                    }                                                                         \
                }                                                                             \
            }                                                                                 \
        }                                                                                     \
        const uint64_t _m_dead = GBALLOT(_dead);                                               \
        SG_COUNT(0, 1);                                                                       \
        FOR_GLANES                                                                             \
        {                                                                                     \
            /* lanes without a live candidate read the start of the ring and are masked */    \
            const int live = LV(_alive);                                                      \
            const uint32_t r0 = live ? lz_ridx<L>(st, LV(_q)) : 0u;                           \
            const uint32_t r1 = live ? lz_ridx<L>(st, LV(_q) + best - 1) : 0u;                \
            const uint32_t w0 = lds_u32(lds->ring, r0);                                       \
            const uint32_t g1 = lds_u32(lds->ring, r1) & 0xffffu;                             \
            LV(_w0) = w0;                                                                     \
            LV(_pass) = live && g1 == sb && (w0 & 0xffffu) == (s0123 & 0xffffu);              \
            LV(_maybe) = live && w0 == s0123;                                                 \
        }                                                                                     \
        uint64_t _todo = GBALLOT(_pass);                                                       \
        const uint64_t _m_maybe = cap > 3u ? GBALLOT(_maybe) : 0ull;                           \
        while (_todo != 0) {                                                                  \
            SG_COUNT(6, 1);                                                                   \
            /* passers that can beat best_len: any at all while best_len is 2, later only     \
             * those whose first four bytes match */                                          \
            const uint64_t _cand = best >= 3u ? (_todo & _m_maybe) : _todo;                   \
            const int _j = _cand ? CTZ64(_cand) : 64; /* (64: none; a group has fewer lanes) */                                         \
            const uint64_t _below = _j < 64 ? (_todo & ((1ull << _j) - 1ull)) : _todo;        \
            const uint32_t _nb = (uint32_t)POPC64(_below);                                    \
            if (_nb >= budget) {                                                              \
                fin = 1; /* the chain budget ends on a candidate that changes nothing */      \
                break;                                                                        \
            }                                                                                 \
            budget -= _nb;                                                                    \
            if (_j == 64)                                                                     \
                break;                                                                        \
            const uint32_t _qj = GREADLANE(_q, _j);                                            \
            uint32_t _len = 3;                                                                \
            if ((_m_maybe >> _j) & 1ull)                                                      \
                SG_LCP(_qj, _len);                                                            \
            int _improved = 0;                                                                \
            if (_len > best) {                                                                \
                where = _qj;                                                                  \
                best = _len;                                                                  \
                _improved = 1;                                                                \
                if (_len >= nice) {                                                           \
                    fin = 1;                                                                  \
                    break;                                                                    \
                }                                                                             \
            }                                                                                 \
            if (--budget == 0) {                                                              \
                fin = 1;                                                                      \
                break;                                                                        \
            }                                                                                 \
            if (_improved) {                                                                  \
                SG_PEEKP(best - 1u, sb);                                                  \
                sb &= 0xffffu;                                                                \
                FOR_GLANES                                                                     \
                {                                                                             \
                    const int live = LV(_alive) && GLANE > _j &&                               \
                                     (LV(_w0) & 0xffffu) == (s0123 & 0xffffu);                \
                    const uint32_t r1 = live ? lz_ridx<L>(st, LV(_q) + best - 1) : 0u;        \
                    LV(_pass) = live && (lds_u32(lds->ring, r1) & 0xffffu) == sb;            \
                }                                                                             \
                _todo = GBALLOT(_pass);                                                        \
            } else {                                                                          \
                _todo &= ~((2ull << _j) - 1ull);                                              \
            }                                                                                 \
        }                                                                                     \
        if (_m_dead != 0 && !fin)                                                             \
            fin = 1; /* the chain leaves the window (:1519) */                                \
    } while (0)

/* Lanes with ACT set hold a candidate (ring index R) whose first four bytes are p's: every such lane
 * compares on, four bytes a step against the string at p (pv), for at most KLIM dwords in all.
 * LEN becomes the length of the common prefix (possibly beyond cap: the caller cuts it); a lane
 * still equal after 4 KLIM bytes keeps ACT set and LEN at that lower bound. */
#define SG_LANE_LCP(ACT, R, LEN, KLIM)                                                        \
    do {                                                                                      \
        uint64_t _am = GBALLOT(ACT);                                                           \
        for (uint32_t _k = 1; _am != 0 && _k < (KLIM) && 4u * _k < cap; _k++) {               \
            const uint32_t _pk = GREADLANE(pv, _k);                                            \
            FOR_GLANES                                                                         \
            {                                                                                 \
                if (LV(ACT)) {                                                                \
                    const uint32_t d = lds_u32(lds->ring, LV(R) + 4u * _k) ^ _pk;             \
                    LV(LEN) = d ? 4u * _k + ((uint32_t)CTZ32(d) >> 3) : 4u * _k + 4u;         \
                    LV(ACT) = d == 0u;                                                        \
                }                                                                             \
            }                                                                                 \
            _am = GBALLOT(ACT);                                                                \
        }                                                                                     \
    } while (0)

/* A chain of at most four loads that is shorter than the chain budget -- two searches in three on text;
 * this is one load of it (entries E, number B).
 * The budget cannot run out, so the walk (src/deflate.c:1455-1519) ends in: the first candidate that
 * reaches nice_match if there is one, otherwise the longest one, the nearest of them first.  Every
 * lane works out its own candidate's length, four bytes a step against the string at p (a step is
 * one instruction stream for all 64 of them, where the walk spends a long compare per candidate that
 * passes the pre-check); when one candidate is left after the first eight bytes the whole wave
 * compares the rest of it at once.  Sets best / where; `bail`: two candidates agree with p for more
 * than 4 GRP bytes, the walk below takes over. */
#define SG_EVAL_ONE(E, B)                                                                        \
    do {                                                                                      \
        LANEVAR(uint32_t, _q);                                                                \
        LANEVAR(uint32_t, _r);                                                                \
        LANEVAR(uint32_t, _len);                                                              \
        LANEVAR(int, _act);                                                                   \
        SG_COUNT(12, 1);                                                                      \
        SG_PV_LOAD();                                                                         \
        FOR_GLANES                                                                             \
        {                                                                                     \
            const uint32_t _v = (B)*GRP + (uint32_t)GLANE;                                     \
            const uint32_t q = tileA + (LV(E) & ZD_TILE_MASK) - (_v < nA ? 0u : ZD_TILE);     \
            const int live = _v < total && q > (_v == 0u ? hfloor : floor_pos);               \
            const uint32_t r = live ? lz_ridx<L>(st, q) : 0u;                                 \
            const uint32_t x = lds_u32(lds->ring, r) ^ s0123;                                 \
            const int m3 = live && (x & 0xffffffu) == 0u;                                     \
            LV(_q) = q;                                                                       \
            LV(_r) = r;                                                                       \
            LV(_len) = m3 ? (x == 0u ? 4u : 3u) : 0u;                                         \
            LV(_act) = m3 && x == 0u;                                                         \
        }                                                                                     \
        uint64_t _am = GBALLOT(_act);                                                          \
        uint32_t _k = 1;                                                                      \
        while (_am != 0 && 4u * _k < cap) {                                                   \
            if (_k >= GRP) {                                                                  \
                bail = 1;                                                                     \
                break;                                                                        \
            }                                                                                 \
            if (_k >= 2u && (_am & (_am - 1ull)) == 0) {                                      \
                const int _j = CTZ64(_am);                                                    \
                const uint32_t _qj = GREADLANE(_q, _j);                                        \
                uint32_t _l;                                                                  \
                SG_LCP(_qj, _l);                                                              \
                FOR_GLANES                                                                     \
                {                                                                             \
                    if (GLANE == _j)                                                           \
                        LV(_len) = _l;                                                        \
                }                                                                             \
                break;                                                                        \
            }                                                                                 \
            const uint32_t _pk = GREADLANE(pv, _k);                                            \
            FOR_GLANES                                                                         \
            {                                                                                 \
                if (LV(_act)) {                                                               \
                    const uint32_t d = lds_u32(lds->ring, LV(_r) + 4u * _k) ^ _pk;            \
                    LV(_len) = d ? 4u * _k + ((uint32_t)CTZ32(d) >> 3) : 4u * _k + 4u;        \
                    LV(_act) = d == 0u;                                                       \
                }                                                                             \
            }                                                                                 \
            _am = GBALLOT(_act);                                                               \
            _k++;                                                                             \
        }                                                                                     \
        if (!bail) {                                                                          \
            LANEVAR(uint32_t, _key);                                                          \
            LANEVAR(int, _nice);                                                              \
            FOR_GLANES                                                                         \
            {                                                                                 \
                const uint32_t l = LV(_len) < cap ? LV(_len) : cap;                           \
                LV(_nice) = l >= nice;                                                        \
                LV(_key) = ((511u - l) << 8) | (uint32_t)GLANE; /* longest first, then nearest */ \
            }                                                                                 \
            const uint64_t _mn = GBALLOT(_nice);                                               \
            const uint32_t _kk = _mn ? GREADLANE(_key, CTZ64(_mn)) : GMIN_U32(_key);           \
            const uint32_t _l = 511u - (_kk >> 8);                                            \
            if (_l > best) {                                                                  \
                best = _l;                                                                    \
                where = GREADLANE(_q, _kk & 0xffu);                                            \
            }                                                                                 \
        }                                                                                     \
    } while (0)

/* four window positions at once: which of the 16-bit strings at byte offsets 0..3 of the
 * dword pair (LO, HI) equal the pre-check pair */
#define SG_M4(LO, HI, K)                                                                      \
    do {                                                                                      \
        const uint32_t _a = (LO) ^ _sb2, _b = (((LO) >> 8) | ((HI) << 24)) ^ _sb2;            \
        _m |= ((uint32_t)((_a & 0xffffu) == 0) | ((uint32_t)((_b & 0xffffu) == 0) << 1) |     \
               ((uint32_t)((_a >> 16) == 0) << 2) | ((uint32_t)((_b >> 16) == 0) << 3))       \
              << (4 * (K));                                                                   \
    } while (0)

/* The same search for a position whose chain is long: instead of following the chain,
 * sweep the window itself, newest position first, 1024 positions per step, for the two
 * bytes the pre-check wants to see at best_len-1 (:1462-1465).  A candidate that fails
 * the pre-check costs the reference nothing -- no chain budget, no effect on best_len --
 * so only positions that show those two bytes AND start with the same three bytes as p
 * (what being on p's chain and passing :1466-1467 amounts to) matter, in the same
 * newest-first order.  A dense bucket costs a walk 64 candidates per step; the sweep
 * covers them at 1024 positions per step without loading a single chain entry. */
#define SG_SWEEP(Q0)                                                                          \
    do {                                                                                      \
        uint32_t _qn = (Q0);                    /* newest position not looked at yet */      \
        uint32_t _qlo = floor_pos + 1u;         /* oldest live position (:1519) */            \
        if (job.cfg.hbits != 15u && floor_pos > st.base && p - floor_pos == job.cfg.max_dist) { \
            int _blk; /* the position at exactly MAX_DIST counts if it heads the reference's chain */ \
            SG_HEAD_BLOCKED(floor_pos, p, LZ_MEMB_ALL, _blk);                                 \
            if (!_blk)                                                                        \
                _qlo = floor_pos;                                                             \
        }                                                                                     \
        while (!fin && _qn >= _qlo) {                                                         \
            const uint32_t _off = best - 1u;                                                  \
            const uint32_t _rtop = _qn + _off, _rlo = _qlo + _off;                            \
            const uint32_t _R0 = _rtop & ~(16u * GRP - 1u);                                              \
            const uint32_t _sb2 = sb * 0x10001u;                                              \
            LANEVAR(uint32_t, _m16);                                                          \
            LANEVAR(int, _has);                                                               \
            SG_COUNT(0, 1);                                                                   \
            FOR_GLANES                                                                         \
            {                                                                                 \
                const uint32_t _x = _R0 + 16u * (uint32_t)GLANE;                               \
                uint32_t _m = 0;                                                              \
                if (_x <= _rtop && _x + 16u > _rlo) {                                         \
                    const uint8_t *_src = &lds->ring[lz_ridx<L>(st, _x)];                     \
                    const uint32_t _w0 = ld_u32(_src), _w1 = ld_u32(_src + 4),                \
                                   _w2 = ld_u32(_src + 8), _w3 = ld_u32(_src + 12),           \
                                   _w4 = ld_u32(_src + 16);                                   \
                    SG_M4(_w0, _w1, 0);                                                       \
                    SG_M4(_w1, _w2, 1);                                                       \
                    SG_M4(_w2, _w3, 2);                                                       \
                    SG_M4(_w3, _w4, 3);                                                       \
                    const uint32_t _hj = _rtop - _x >= 15u ? 15u : _rtop - _x;                \
                    const uint32_t _lj = _rlo > _x ? _rlo - _x : 0u;                          \
                    _m &= ((2u << _hj) - 1u) & ~((1u << _lj) - 1u);                           \
                    /* of those, the ones that start with p's three bytes (every lane goes through \
                     * its own few at the same time; the wave then visits what is left, in order); \
                     * not where they come thick -- a run of one byte: they all do */            \
                    uint32_t _left = POPC64((uint64_t)_m) <= 4 ? _m : 0u;                     \
                    while (_left != 0) {                                                      \
                        const uint32_t _b = (uint32_t)CTZ32(_left);                           \
                        _left &= _left - 1u;                                                  \
                        const uint32_t _t3 = lds_u32(lds->ring, lz_ridx<L>(st, _x + _b - _off)); \
                        if (((_t3 ^ s0123) & 0xffffffu) != 0)                                 \
                            _m &= ~(1u << _b);                                                \
                    }                                                                         \
                }                                                                             \
                LV(_m16) = _m;                                                                \
                LV(_has) = _m != 0;                                                           \
            }                                                                                 \
            uint64_t _any = GBALLOT(_has);                                                     \
            int _moved = 0;                                                                   \
            while (_any != 0 && !_moved) {                                                    \
                const int _l = 63 - CLZ64(_any);                                              \
                uint32_t _mm = GREADLANE(_m16, _l);                                            \
                while (_mm != 0) {                                                            \
                    const int _j = 31 - CLZ32(_mm);                                           \
                    _mm &= ~(1u << _j);                                                       \
                    const uint32_t _qj = _R0 + 16u * (uint32_t)_l + (uint32_t)_j - _off;      \
                    const uint32_t _t = GUNI(lds_u32(lds->ring, lz_ridx<L>(st, _qj)));         \
                    if (((_t ^ s0123) & 0xffffffu) != 0)                                      \
                        continue; /* not on p's chain, or fails :1466-1467 */                 \
                    uint32_t _len = 3;                                                        \
                    if (cap > 3u && _t == s0123)                                              \
                        SG_LCP(_qj, _len);                                                    \
                    if (_len > best) {                                                        \
                        where = _qj;                                                          \
                        best = _len;                                                          \
                        _moved = 1;                                                           \
                        if (_len >= nice) {                                                   \
                            fin = 1;                                                          \
                            break;                                                            \
                        }                                                                     \
                    }                                                                         \
                    if (--budget == 0) {                                                      \
                        fin = 1;                                                              \
                        break;                                                                \
                    }                                                                         \
                    if (_moved) {                                                             \
                        /* new best_len: other bytes at another offset from here on */        \
                        SG_PEEKP(best - 1u, sb);                                          \
                        sb &= 0xffffu;                                                        \
                        _qn = _qj - 1u;                                                       \
                        break;                                                                \
                    }                                                                         \
                }                                                                             \
                if (fin)                                                                      \
                    break;                                                                    \
                _any &= ~(1ull << _l);                                                        \
            }                                                                                 \
            if (fin || _moved)                                                                \
                continue;                                                                     \
            if (_R0 <= _rlo)                                                                  \
                break; /* the sweep reached the far end of the window */                      \
            _qn = _R0 - 1u - _off;                                                            \
        }                                                                                     \
    } while (0)

/* ---- hops: the parse as a walk over the match table ---------------------------------------
 *
 * With the table (match_table.h) a parser that stands at x with no match pending does not have
 * to search: r2[x] says what longest_match finds there, rl[x+1] whether the lazy evaluation at
 * x+1 finds something longer, and so on -- everything deflate_slow does until it next stands at
 * a position with no match pending follows from a few table entries.  Each lane works that out
 * for one of 64 consecutive positions at once (a HOP: the literals and the match that are
 * emitted, the position the parse stands at next); the wave then only follows the hops.  A
 * position whose entries are not all known (MT_INCOMPLETE, or a fourth lazy step) has no hop
 * and goes through the search below as before. */
#define SGH_VALID 0x80000000u
#define SGH_MATCH 0x40000000u /* literals (bits 28-29 say how many), then the match in bits 0-23 (MT_LEN / MT_DIST) */

/* longest_match at position X for prev_length KEY, given A = r2[X]: the same match if it is longer,
 * none otherwise -- where the entry says that holds (MT_RLOK) */
#define SG_RL(X, A, KEY) (((A)&MT_RLOK) ? (MT_LEN(A) > (KEY) ? (A) : MT_NONE) : MT_INCOMPLETE)

/* hops of positions X0 .. X0+GRP-1: LV(hop), and the four input bytes at each position LV(hby) */
#define SG_HOP_LOAD(X0)                                                                       \
    FOR_GLANES                                                                                 \
    {                                                                                         \
        const uint32_t x = (X0) + (uint32_t)GLANE;                                             \
        uint32_t h = 0, by = 0;                                                               \
        if ((uint64_t)x + 4u <= job.n) {                                                      \
            const uint32_t a0 = job.r2[x];                                                    \
            by = ld_u32(job.in + x);                                                          \
            if (!(a0 & MT_INCOMPLETE)) {                                                      \
                const uint32_t l1 = MT_LEN(a0);                                               \
                if (l1 == 2u) {                                                               \
                    h = SGH_VALID;                                                            \
                } else {                                                                      \
                    uint32_t fin = a0, nlit = 0;                                              \
                    int good = 1;                                                             \
                    if (l1 < job.cfg.lazy) {                                                  \
                        const uint32_t a1 = job.r2[x + 1u];                                   \
                        const uint32_t b1 = SG_RL(x + 1u, a1, l1);                            \
                        if (b1 & MT_INCOMPLETE) {                                             \
                            good = 0;                                                         \
                        } else if (MT_LEN(b1) > l1) {                                         \
                            const uint32_t l2 = MT_LEN(b1);                                   \
                            fin = b1;                                                         \
                            nlit = 1;                                                         \
                            if (l2 < job.cfg.lazy) {                                          \
                                /* rl[x+2] answers for the length in r2[x+1]: is that l2? */   \
                                const uint32_t a2 = job.r2[x + 2u];                           \
                                const uint32_t b2 = SG_RL(x + 2u, a2, l2);                    \
                                if (((a1 | b2) & MT_INCOMPLETE) || MT_LEN(a1) != l2) {        \
                                    good = 0;                                                 \
                                } else if (MT_LEN(b2) > l2) {                                 \
                                    const uint32_t l3 = MT_LEN(b2);                           \
                                    fin = b2;                                                 \
                                    nlit = 2;                                                 \
                                    if (l3 < job.cfg.lazy) {                                  \
                                        const uint32_t a3 = job.r2[x + 3u];                   \
                                        const uint32_t b3 = SG_RL(x + 3u, a3, l3);            \
                                        if (((a2 | b3) & MT_INCOMPLETE) || MT_LEN(a2) != l3 || MT_LEN(b3) > l3) \
                                            good = 0;                                         \
                                    }                                                         \
                                }                                                             \
                            }                                                                 \
                        }                                                                     \
                    }                                                                         \
                    if (good)                                                                 \
                        h = SGH_VALID | SGH_MATCH | (nlit << 28) | (fin & 0xffffffu);         \
                }                                                                             \
            }                                                                                 \
        }                                                                                     \
        LV(hop) = h;                                                                          \
        LV(hby) = by;                                                                         \
        LV(hr2) = (uint64_t)x + 4u <= job.n ? job.r2[x] : MT_INCOMPLETE;                      \
    }

/* What a parser records of the positions of its own segment that it visits with no match pending
 * lives in one register, a lane per position of the current block of GRP: the token count there,
 * bit 16 "visited fresh", bit 17 "a literal was owed".  When the parse leaves the block the counts go
 * to sidx, and two ballots make the block's words of the trace bitmaps -- kinds first: whoever sees a
 * bit of a trace word finds its kind in place. */
#define SG_TRACE_FLUSH()                                                                      \
    do {                                                                                      \
        LANEVAR(int, _fr);                                                                    \
        LANEVAR(int, _kd);                                                                    \
        FOR_GLANES                                                                             \
        {                                                                                     \
            sidx[sd_blk * GRP + (uint32_t)GLANE] = (uint16_t)LV(sdx);                          \
            LV(_fr) = (int)((LV(sdx) >> 16) & 1u);                                            \
            LV(_kd) = (int)((LV(sdx) >> 17) & 1u);                                            \
        }                                                                                     \
        const uint64_t _mf = GBALLOT(_fr), _mk = GBALLOT(_kd);                                  \
        ON_GLANE0                                                                              \
        {                                                                                     \
            if (GRP >= 32u) {                                                                 \
                for (uint32_t _j = 0; _j < GRP / 32u; _j++) {                                 \
                    const uint32_t _w = sd_blk * (GRP / 32u) + _j;                            \
                    lds->tkind[s][_w] = (uint32_t)(_mk >> (32u * _j));                        \
                    LDS_STORE_REL(&lds->trace[s][_w], (uint32_t)(_mf >> (32u * _j)));         \
                }                                                                             \
            } else { /* (narrow groups: part of a word; this parser is the word's only writer) */ \
                const uint32_t _w = sd_blk * GRP / 32u, _sh = sd_blk * GRP % 32u;             \
                const uint32_t _m = ((1u << (GRP & 31u)) - 1u) << _sh;                        \
                lds->tkind[s][_w] = (lds->tkind[s][_w] & ~_m) | ((uint32_t)_mk << _sh);       \
                LDS_STORE_REL(&lds->trace[s][_w], (lds->trace[s][_w] & ~_m) | ((uint32_t)_mf << _sh)); \
            }                                                                                 \
        }                                                                                     \
    } while (0)

/* one token to the segment's token area (through a register, 64 at a time) */
#define SG_EMIT(SYM)                                                                          \
    do {                                                                                      \
        const uint32_t _sym = (SYM);                                                          \
        FOR_GLANES                                                                             \
        {                                                                                     \
            if ((uint32_t)GLANE == nstaged)                                                    \
                LV(stg) = _sym;                                                               \
        }                                                                                     \
        nstaged++;                                                                            \
        ntok++;                                                                               \
        if (nstaged == GRP) {                                                                 \
            FOR_GLANES { tok[ntok - GRP + (uint32_t)GLANE] = LV(stg); }                        \
            nstaged = 0;                                                                      \
        }                                                                                     \
    } while (0)

/* parse from (p, cur_len, cur_at, pending) -- a state of the serial parse, or the fresh
 * state at the start of segment s -- until the parse can be handed to a later segment's
 * tokens, gives up, or leaves the super-step */
template <bool TABLE> /* the buffer has a match table (job.r2): a build without it carries none of its code */
DEV void sg_parse_segment(const LzJob &job, SgLds *lds, const SgScratch &scr, uint32_t s,
                          uint32_t p, uint32_t cur_len, uint32_t cur_at, int pending)
{
    typedef SgLds L;
    LzState st;
    st.lo = GUNI(lds->lo);
    st.hi = GUNI(lds->hi);
    st.wrap_base = GUNI(lds->wrap_base);
    st.nsyms = st.nstaged = st.nblocks = st.blk_sym0 = st.blk_in0 = st.pr_hi = 0;
    st.n = job.n;
    st.si = 0;
    st.it = 0;

    const uint32_t S0 = GUNI(lds->S0);
    const uint64_t E64 = (uint64_t)S0 + SG_SPAN;
    const uint32_t E = E64 < job.n ? (uint32_t)E64 : job.n; /* end of the super-step */
    const uint32_t a_s = S0 + s * SG_G;
    const uint32_t e_s = a_s + SG_G < E ? a_s + SG_G : E;

    st.base = sg_base(job.cfg, p, job.n);
    {
        uint64_t end = (uint64_t)st.base + 2ull * job.cfg.wsize;
        st.data_end = end < job.n ? (uint32_t)end : job.n;
    }
    uint32_t *tok = scr.tok + s * SG_TOKCAP;
    uint16_t *sidx = scr.sidx + s * SG_TRACE;
    uint32_t ntok = 0, nstaged = 0, exit_kind = 0;

    /* everything a search needs from memory that does not depend on the candidates is
     * kept in registers, one value per lane, and read with v_readlane:
     *   mrk/mcn  rank|hib and cnt of 64 consecutive positions
     *   pw       256 bytes of the window around p
     *   pv       the string at p as the long compare wants it (dword l = bytes 4l..4l+3)
     *   stg      tokens not yet written out;  sdx  token counts and kinds at fresh positions (SG_TRACE_FLUSH) */
    LANEVAR(uint32_t, mha); /* of 64 consecutive positions: rank - 1 (the newest older entry of the own tile's run), */
    LANEVAR(uint32_t, mhb); /* ... hib, */
    LANEVAR(uint32_t, mna); /* ... the number of older entries in the own tile */
    LANEVAR(uint32_t, mto); /* ... and in both tiles: four v_readlane and no unpacking per search */
    LANEVAR(uint32_t, mfl); /* ... the oldest position a link of its chain may have (:1519), minus one */
    LANEVAR(uint32_t, mhf); /* ... and the same for the chain head, which may lie at exactly MAX_DIST (:2032) */
    LANEVAR(uint32_t, mby); /* ... its four bytes of the window */
    LANEVAR(uint32_t, mcp); /* ... and how long a match may get there and which length ends a search: lookahead and */
    LANEVAR(uint32_t, mni); /*     nice_match, cut to what is left of the window's data (:1430-1436) */
    LANEVAR(uint32_t, pv);
    LANEVAR(uint32_t, stg);
    LANEVAR(uint32_t, sdx);
    LANEVAR(uint32_t, hop); /* hops of the positions from hop_at on, and the input bytes there */
    LANEVAR(uint32_t, hby);
    LANEVAR(uint32_t, hr2); /* ... and their table entries themselves */
    FOR_GLANES { LV(mha) = LV(mhb) = LV(mna) = LV(mto) = LV(mfl) = LV(mhf) = LV(mby) = LV(mcp) = LV(mni) = LV(pv) = LV(stg) = LV(sdx) = LV(hop) = LV(hby) = LV(hr2) = 0; }
    uint32_t hop_at = p + 4096u; /* (out of range, as the other caches) */
    /* the two caches start out of range of p, so that their one range test fails */
    uint32_t mt_at = p + 4096u, pv_at = 0xffffffffu;
    uint32_t lit = 0;                      /* the byte at p-1 */
    if (pending)
        lit = GUNI(lds->ring[lz_ridx<L>(st, p - 1u)]);
    uint32_t sd_blk = 0xffffffffu;         /* block of GRP positions sdx belongs to */

    for (;;) {
        uint32_t look = st.data_end - p;
        if (look < ZD_MIN_LOOKAHEAD) {
            const uint32_t base0 = st.base, end0 = st.data_end;
            lz_refill(job, st, p);
            if (st.base != base0 || st.data_end != end0)
                mt_at = p + 4096u; /* (the window has moved: the register cache holds values that depend on it) */
            look = st.data_end - p;
            if (look == 0) {
                exit_kind = SG_EXIT_END;
                break;
            }
        }
        SG_COUNT(4, 1);
        /* no match pending: the state is (p, pending) alone -- fresh or neutral */
        const int fresh = cur_len == 2u;
        if (p >= e_s) {
            if (p >= E) {
                exit_kind = SG_EXIT_LAST;
                break;
            }
            if (fresh) {
                /* the segment p lies in recorded the positions its own parser was fresh at */
                const uint32_t t = (p - S0) / SG_G, r = (p - S0) % SG_G;
                /* (the other parser may still be running: its words are published kinds first) */
                const uint32_t tword = GUNI(LDS_LOAD_ACQ(&lds->trace[t][r >> 5]));
                if (((tword >> (r & 31u)) & 1u) &&
                    ((GUNI(lds->tkind[t][r >> 5]) >> (r & 31u)) & 1u) == (uint32_t)pending) {
                    exit_kind = SG_EXIT_SYNCED;
                    break;
                }
            }
            if (p >= e_s + SG_OV) {
                exit_kind = SG_EXIT_UNSYNCED;
                break;
            }
        } else if (fresh) { /* p >= a_s always: every parser starts inside its segment */
            const uint32_t r = p - a_s;
            if (r / GRP != sd_blk) {
                if (sd_blk != 0xffffffffu)
                    SG_TRACE_FLUSH();
                sd_blk = r / GRP;
                FOR_GLANES { LV(sdx) = 0; }
            }
            const uint32_t rec = ntok | 0x10000u | ((uint32_t)pending << 17);
            FOR_GLANES
            {
                if ((uint32_t)GLANE == r % GRP)
                    LV(sdx) = rec;
            }
        }

        if (TABLE && p - hop_at >= GRP) {
            hop_at = p;
            SG_HOP_LOAD(p);
        }
        if (TABLE && fresh) {
            const uint32_t h = GREADLANE(hop, p - hop_at);
            if (h & SGH_VALID) {
                const uint32_t by = GREADLANE(hby, p - hop_at);
                SG_COUNT(7, 1);
                if (pending)
                    SG_EMIT(lit);
                if (h & SGH_MATCH) {
                    const uint32_t nlit = (h >> 28) & 3u;
                    for (uint32_t i = 0; i < nlit; i++)
                        SG_EMIT((by >> (8u * i)) & 0xffu);
                    SG_EMIT((MT_DIST(h) << 16) | (MT_LEN(h) - 3u));
                    p += nlit + MT_LEN(h);
                    pending = 0;
                } else {
                    lit = by & 0xffu;
                    pending = 1;
                    p++;
                }
                continue;
            }
        }
        /* what a loop top and a search need to know of p is kept for 64 consecutive positions, a lane each */
        if (p - mt_at >= GRP - SG_PICK_AHEAD) { /* (and the positions a search may choose its chain from) */
            mt_at = p;
            FOR_GLANES
            {
                const uint32_t x = p + (uint32_t)GLANE;
                const int ok = x + 2 < job.n;
                uint32_t rh_x = 0, cn_x = 0;
                if (job.dir) {
                    if (ok)
                        sg_link<L>(job, lds, st, x, (uint32_t)job.rank[x], rh_x, cn_x);
                } else {
                    rh_x = ok ? ((uint32_t)job.rank[x] | ((uint32_t)job.hib[x] << 16)) : 0u;
                    cn_x = ok ? job.cnt[x] : 0u;
                }
                LV(mha) = (rh_x & 0xffffu) - 1u;
                LV(mhb) = rh_x >> 16;
                LV(mna) = cn_x & 0xffffu;
                LV(mto) = (cn_x & 0xffffu) + (cn_x >> 16);
                const uint32_t far_x = x - st.base > job.cfg.max_dist;
                LV(mfl) = far_x ? x - job.cfg.max_dist : st.base;
                LV(mhf) = LV(mfl) - far_x;
                LV(mby) = lds_u32(lds->ring, lz_ridx<L>(st, x)); /* (what lies behind the ring's data is never used) */
                const uint32_t look_x = st.data_end - x;
                LV(mcp) = look_x < 258u ? look_x : 258u;
                LV(mni) = job.cfg.nice < look_x ? job.cfg.nice : look_x;
            }
        }
        const uint32_t s0123 = GREADLANE(mby, p - mt_at);
        const uint32_t prev_len = cur_len, prev_at = cur_at;
        cur_len = 2;
        /* no hop, but the table may still know this one search: the entry itself when no match is
         * pending, or -- where it says so -- what it implies for a longer prev_length */
        int known = 0;
        if (TABLE && look >= 3 && prev_len < job.cfg.lazy) {
            const uint32_t a = GREADLANE(hr2, p - hop_at);
            if (prev_len == 2u ? !(a & MT_INCOMPLETE) : (a & MT_RLOK) != 0u) {
                known = 1;
                SG_COUNT(10, 1);
                if (MT_LEN(a) > prev_len) {
                    cur_len = MT_LEN(a);
                    cur_at = p - MT_DIST(a);
                }
            }
        }
        /* (look >= 3 need not be asked: within MIN_LOOKAHEAD of the end of the window's data fill_window has run,
         * so fewer than three bytes are only left at the end of the input, where positions own no string and
         * their chains are empty: total == 0 below) */
        if (prev_len < job.cfg.lazy && !known) {
            const uint32_t rp = lz_ridx<L>(st, p);
            const uint32_t nA = GREADLANE(mna, p - mt_at), total = GREADLANE(mto, p - mt_at);
            /* A match pending: anything longer shares prev_len + 1 bytes with p, so every trigram in
             * them has been seen before -- if one of them (of those whose chain lengths are in the
             * register cache) has an EMPTY chain, there is nothing longer and no need to look. */
            int none_longer = 0;
            if (SG_EMPTY_SKIP && total != 0 && prev_len >= 3u) {
                LANEVAR(int, empty);
                const uint32_t l0 = p - mt_at;
                FOR_GLANES
                {
                    const uint32_t l = (uint32_t)GLANE;
                    LV(empty) = l > l0 && l <= l0 + (prev_len - 2u) && LV(mto) == 0u && (uint64_t)mt_at + l + 3u <= job.n;
                }
                none_longer = GBALLOT(empty) != 0;
                SG_COUNT(9, (unsigned)none_longer);
            }
            if (total != 0 && !none_longer) {
                SG_COUNT(5, 1);
                const int32_t hiA = (int32_t)GREADLANE(mha, p - mt_at), hiB = (int32_t)GREADLANE(mhb, p - mt_at);
                const uint32_t tileA = p & ~ZD_TILE_MASK;
                uint32_t floor_pos = GREADLANE(mfl, p - mt_at);
                const uint32_t cap = GREADLANE(mcp, p - mt_at), nice = GREADLANE(mni, p - mt_at);
                uint32_t best = prev_len, where = cur_at, sb = 0;
                uint32_t budget = prev_len >= job.cfg.good ? job.cfg.chain >> 2 : job.cfg.chain;
                int fin = 0, head_seen = 0;
                LANEVAR(uint32_t, e0);
                LANEVAR(uint32_t, e1);
                LANEVAR(uint32_t, e2);
                LANEVAR(uint32_t, e3);
                LANEVAR(uint32_t, f0);
                LANEVAR(uint32_t, f1);
                LANEVAR(uint32_t, f2);
                LANEVAR(uint32_t, f3);
                SG_LOAD(e0, 0u);
                int bail = 0, searched = 0;
                /* (one comparison: the chain is shorter than the budget and than four loads.  With no room for a
                 * longer match -- best >= look, at the very end of the data -- no candidate gets past cap <= best
                 * and the result is the same as that of not searching: min(best, look)) */
                const uint32_t one_lim = budget < 4u * GRP + 1u ? budget : 4u * GRP + 1u;
                if (SG_ONE && SG_ONE_ON && job.cfg.hbits == 15u && total < one_lim) {
                    if (total > GRP) {
                        SG_LOAD(e1, 1u);
                        if (total > 2u * GRP) {
                            SG_LOAD(e2, 2u);
                            SG_LOAD(e3, 3u);
                        }
                    }
                    /* The chain head may lie at exactly MAX_DIST (:2032), later links may not (:1519): q0 > base
                     * and p - q0 <= MAX_DIST is q0 > floor_pos - far, and lane 0 tests its candidate against that.
                     * A head that fails it has nothing but older entries behind it, which fail theirs: the search
                     * finds nothing, which is what not calling longest_match comes to (match_length <= prev_length
                     * either way). */
                    const uint32_t hfloor = GREADLANE(mhf, p - mt_at);
                    SG_EVAL_ONE(e0, 0u);
                    if (!bail && best < nice && total > GRP) {
                        SG_EVAL_ONE(e1, 1u);
                        if (!bail && best < nice && total > 2u * GRP) {
                            SG_EVAL_ONE(e2, 2u);
                            if (!bail && best < nice && total > 3u * GRP)
                                SG_EVAL_ONE(e3, 3u);
                        }
                    }
                    head_seen = !bail;
                    if (bail) {
                        best = prev_len;
                        where = cur_at;
                    }
                    searched = !bail;
                }
                if (!searched) {
                /* a long chain is searched as a staircase over shorter ones (below); a short one is
                 * walked as the reference walks it, 64 candidates a step */
                const int stair = SG_STAIR && job.cfg.hbits == 15u && total >= job.stair_min;
                if (!stair && total > GRP) { /* most chains of text fit one load */
                    SG_LOAD(e1, 1u);
                    SG_LOAD(e2, 2u);
                    SG_LOAD(e3, 3u);
                }
                /* the chain head may lie at exactly MAX_DIST (:2032), later links may not */
                const uint32_t ent0 = GREADLANE(e0, 0);
                const uint32_t q0 = tileA + (ent0 & ZD_TILE_MASK) - (nA ? 0u : ZD_TILE);
                int head_ok = q0 > st.base && p - q0 <= job.cfg.max_dist;
                if (head_ok && p - q0 == job.cfg.max_dist && job.cfg.hbits != 15u) {
                    int blocked;
                    SG_HEAD_BLOCKED(q0, p, LZ_MEMB_ALL, blocked);
                    head_ok = !blocked;
                }
                if (!head_ok) {
                    fin = 2; /* no chain head in the window: longest_match is not called */
                } else {
                    head_seen = 1;
                    if (best >= look)
                        fin = 1;
                }
                if (!fin) {
                    if (best <= 3u) /* the two bytes are among the four read at p */
                        sb = (s0123 >> (8u * (best - 1u))) & 0xffffu;
                    else {
                        SG_PEEKP(best - 1u, sb);
                        sb &= 0xffffu;
                    }
                }
                if (!fin && stair) {
                    /* ---- the search as a staircase (match_table.h has the argument) ----------------
                     * A candidate that fails the pre-check is free, so while the chain budget lasts the
                     * walk's result is: the nearest candidate longer than best, then the nearest one
                     * beyond it that is longer still, ...  A candidate longer than best shares best + 1
                     * bytes with p and so lies on the chain of every position p + j, j <= best - 2,
                     * shifted by j: each step walks the SHORTEST of those chains (of the ones whose
                     * lengths are in the register cache) instead of p's own.  Whether the budget could
                     * have ended the reference's walk before the last step is checked afterwards; if it
                     * could, the walk is done the reference's way, down to the last step's candidate. */
                    const uint32_t budget0 = budget;
                    uint32_t bnd = p, nrec = 0, qs = best >= 5u ? p : 0xffffffffu;
                    int more = 1;
                    while (more) {
                        more = 0;
                        uint32_t j = 0, nAj = 0, totj = 0;
                        int32_t hiAj = 0, hiBj = 0;
                        if (best >= 3u) {
                            /* the shortest chain among those whose lengths are in the register cache */
                            LANEVAR(uint32_t, tkey);
                            LANEVAR(int, ismin);
                            const uint32_t l0 = p - mt_at, l1 = l0 + (best - 2u);
                            FOR_GLANES
                            {
                                const uint32_t l = (uint32_t)GLANE;
                                LV(tkey) = (l >= l0 && l <= l1 && (uint64_t)mt_at + l + 3u <= job.n) ? LV(mto) : 0xffffffffu;
                            }
                            uint32_t tmin = GMIN_U32(tkey);
                            FOR_GLANES { LV(ismin) = LV(tkey) == tmin; }
                            j = (uint32_t)CTZ64(GBALLOT(ismin)) - l0;
                            /* A long match whose last trigrams lie beyond the cache (a run of equal bytes
                             * and then something else: every chain in reach is the run's own, tens of
                             * thousands of entries): the two chains that end at the byte which has to match
                             * next are worth a trip to memory. */
                            if (tmin > SG_FAR_OVER && l1 >= GRP) {
                                for (uint32_t o = best - 3u; o <= best - 2u; o++) {
                                    if (p + o - mt_at >= GRP && (uint64_t)p + o + 3u <= job.n) {
                                        uint32_t c, rhx;
                                        if (job.dir) {
                                            sg_link<L>(job, lds, st, p + o, (uint32_t)job.rank[p + o], rhx, c);
                                            c = GUNI(c);
                                            rhx = GUNI(rhx);
                                        } else {
                                            c = GUNI(job.cnt[p + o]);
                                            rhx = GUNI((uint32_t)job.rank[p + o]) | (GUNI((uint32_t)job.hib[p + o]) << 16);
                                        }
                                        const uint32_t t = (c & 0xffffu) + (c >> 16);
                                        if (t < tmin) {
                                            tmin = t;
                                            j = o;
                                            nAj = c & 0xffffu;
                                            totj = t;
                                            hiAj = (int32_t)(rhx & 0xffffu) - 1;
                                            hiBj = (int32_t)(rhx >> 16);
                                        }
                                    }
                                }
                            }
                        }
                        if (p + j - mt_at < GRP) {
                            nAj = GREADLANE(mna, p + j - mt_at);
                            totj = GREADLANE(mto, p + j - mt_at);
                            hiAj = (int32_t)GREADLANE(mha, p + j - mt_at);
                            hiBj = (int32_t)GREADLANE(mhb, p + j - mt_at);
                        }
                        const uint32_t tileJ = (p + j) & ~ZD_TILE_MASK;
                        /* A candidate longer than best matches ALL of p's bytes 0 .. best: the four that end
                         * at best (the reference's pre-check looks at the last two of them), the first four,
                         * and four in the middle are looked at before the long compare is spent on it.
                         * (best == 2: the first three bytes are the whole test.) */
                        uint32_t sb4 = 0, smid = 0;
                        const uint32_t omid = best >= 8u ? best / 2u : 0u;
                        if (best >= 3u) {
                            SG_PEEKP(best - 3u, sb4);
                            if (omid)
                                SG_PEEKP(omid, smid);
                        }
                        int found = 0, gone = 0;
                        for (uint32_t bb = 0; bb * GRP < totj && !found && !gone; bb++) {
                            LANEVAR(uint32_t, ej);
                            LANEVAR(uint32_t, _q);
                            LANEVAR(uint32_t, _w0);
                            LANEVAR(uint32_t, _r);
                            LANEVAR(uint32_t, _len);
                            LANEVAR(int, _act);
                            LANEVAR(int, _pass);
                            LANEVAR(int, _gone);
                            if (j == 0u && bb == 0u) {
                                FOR_GLANES { LV(ej) = LV(e0); }
                            } else {
                                FOR_GLANES
                                {
                                    const uint32_t _v = bb * GRP + (uint32_t)GLANE;
                                    int32_t _i = _v < nAj ? hiAj - (int32_t)_v : hiBj - (int32_t)(_v - nAj) - (int32_t)ZD_TILE;
                                    _i = _v < totj ? _i : 0;
                                    LV(ej) = job.sorted[(uint32_t)((int32_t)tileJ + _i)];
                                }
                            }
                            SG_COUNT(0, 1);
                            FOR_GLANES
                            {
                                const uint32_t _v = bb * GRP + (uint32_t)GLANE;
                                const uint32_t qj = tileJ + (LV(ej) & ZD_TILE_MASK) - (_v < nAj ? 0u : ZD_TILE);
                                const uint32_t q = qj - j;
                                const int have = _v < totj;
                                /* the chain leaves the buffer or the window (:1519) */
                                const int g = have && (qj < j || (q < bnd && !(q > floor_pos || q == q0)));
                                /* (an entry newer than the last step's candidate was seen at an earlier level) */
                                const int live = have && !g && q < bnd;
                                const uint32_t r0 = live ? lz_ridx<L>(st, q) : 0u;
                                const uint32_t w0 = lds_u32(lds->ring, r0);
                                int ok = live && ((w0 ^ s0123) & 0xffffffu) == 0u;
                                if (best >= 3u) {
                                    const uint32_t r1 = live ? lz_ridx<L>(st, q + best - 3u) : 0u;
                                    ok = ok && w0 == s0123 && lds_u32(lds->ring, r1) == sb4;
                                    if (omid) {
                                        const uint32_t r2 = live ? lz_ridx<L>(st, q + omid) : 0u;
                                        ok = ok && lds_u32(lds->ring, r2) == smid;
                                    }
                                }
                                LV(_q) = q;
                                LV(_w0) = w0;
                                LV(_r) = r0;
                                LV(_gone) = g;
                                LV(_pass) = ok;
                            }
                            const uint64_t mgone = GBALLOT(_gone);
                            uint64_t todo = GBALLOT(_pass);
                            if (mgone != 0) {
                                todo &= (mgone & (0ull - mgone)) - 1ull; /* what comes after the first such entry is older still */
                                gone = 1;
                            }
                            if (todo != 0) {
                                /* the lengths of all of them at once (the first 32 bytes; what agrees with p
                                 * beyond that gets the wave's long compare when its turn comes), then the steps
                                 * this load holds, one after the other: each the nearest entry longer than best */
                                FOR_GLANES
                                {
                                    const int t = (int)((todo >> GLANE) & 1ull);
                                    const int four = t && cap > 3u && LV(_w0) == s0123;
                                    LV(_len) = t ? (four ? 4u : 3u) : 0u;
                                    LV(_act) = four;
                                }
                                SG_PV_LOAD();
                                SG_LANE_LCP(_act, _r, _len, 8u);
                            }
                            while (todo != 0) {
                                LANEVAR(int, _imp);
                                FOR_GLANES { LV(_imp) = LV(_act) || (LV(_len) < cap ? LV(_len) : cap) > best; }
                                todo &= GBALLOT(_imp);
                                if (todo == 0)
                                    break;
                                const int jl = CTZ64(todo);
                                const uint32_t qx = GREADLANE(_q, jl);
                                uint32_t len = GREADLANE(_len, jl);
                                if ((GBALLOT(_act) >> jl) & 1ull) {
                                    SG_LCP(qx, len);
                                    FOR_GLANES
                                    {
                                        if (GLANE == jl) {
                                            LV(_act) = 0;
                                            LV(_len) = len;
                                        }
                                    }
                                }
                                if (len > cap)
                                    len = cap;
                                if (len > best) {
                                    found = 1;
                                    where = qx;
                                    best = len;
                                    bnd = qx;
                                    nrec++;
                                    if (qs == 0xffffffffu && best >= 5u)
                                        qs = where;
                                    if (best >= nice)
                                        break;
                                }
                                todo &= ~((2ull << jl) - 1ull);
                            }
                        }
                        if (found && best < nice)
                            more = 1;
                    }
                    fin = 1;
                    if (nrec != 0) {
                        /* the budget is charged by the steps, and by candidates that pass the pre-check
                         * without being longer: none while best_len <= 4, so at most as many as p's own
                         * chain has entries between the first step to a level >= 5 and the last step */
                        uint32_t between = 0;
                        if (qs != 0xffffffffu && qs != where) {
                            const uint32_t rk = (uint32_t)hiA + 1u, hb = (uint32_t)hiB;
                            const uint32_t r1 = GUNI((uint32_t)job.rank[where]);
                            const uint32_t newer = (where >> 15) == (p >> 15) ? rk - 1u - r1 : nA + (hb - r1);
                            if (qs == p) {
                                between = newer;
                            } else {
                                const uint32_t r0 = GUNI((uint32_t)job.rank[qs]);
                                between = newer - ((qs >> 15) == (p >> 15) ? rk - 1u - r0 : nA + (hb - r0)) - 1u;
                            }
                        }
                        if (nrec + between >= budget0) {
                            SG_COUNT(8, 1);
                            /* the reference's walk, as far as the last step's candidate: nothing beyond it
                             * is longer, so nothing beyond it changes the result */
                            if (where - 1u > floor_pos)
                                floor_pos = where - 1u;
                            best = prev_len;
                            where = cur_at;
                            budget = budget0;
                            fin = 0;
                            if (best <= 3u)
                                sb = (s0123 >> (8u * (best - 1u))) & 0xffffu;
                            else {
                                SG_PEEKP(best - 1u, sb);
                                sb &= 0xffffu;
                            }
                        }
                    }
                }
                int sweep = 0;
                if (total > 4u * GRP) {
                    /* a sweep step covers 16 GRP window positions, a walk step GRP candidates */
                    if (q0 > floor_pos)
                        sweep = total > 2u * GRP * ((q0 - floor_pos) / (16u * GRP) + 2u);
                }
                if (fin) {
                } else if (sweep) {
                    SG_SWEEP(q0);
                } else {
                    if (stair && total > GRP) {
                        SG_LOAD(e1, 1u);
                        SG_LOAD(e2, 2u);
                        SG_LOAD(e3, 3u);
                    }
                    for (uint32_t b0 = 0;; b0 += 4u) {
                        const int more = (b0 + 4u) * GRP < total;
                        if (more) {
                            /* the next 256 candidates are on their way while these are looked at */
                            SG_LOAD(f0, b0 + 4u);
                            SG_LOAD(f1, b0 + 5u);
                            SG_LOAD(f2, b0 + 6u);
                            SG_LOAD(f3, b0 + 7u);
                        }
                        SG_EVAL(e0, b0);
                        if (!fin && (b0 + 1u) * GRP < total)
                            SG_EVAL(e1, b0 + 1u);
                        if (!fin && (b0 + 2u) * GRP < total)
                            SG_EVAL(e2, b0 + 2u);
                        if (!fin && (b0 + 3u) * GRP < total)
                            SG_EVAL(e3, b0 + 3u);
                        if (fin || !more)
                            break;
                        FOR_GLANES
                        {
                            LV(e0) = LV(f0);
                            LV(e1) = LV(f1);
                            LV(e2) = LV(f2);
                            LV(e3) = LV(f3);
                        }
                    }
                }
                } /* (!searched) */
                if (head_seen) {
                    cur_at = where;
                    cur_len = best < look ? best : look;
                    if (cur_len <= 5) { /* :2038-2047 */
                        if (job.strategy == 1)
                            cur_len = 2;
                        else if (cur_len == 3) {
                            if (p - cur_at > ZD_TOO_FAR)
                                cur_len = 2;
                        }
                    }
                }
            }
        }
        uint32_t sym = 0;
        int emit = 0;
        if (prev_len >= 3 && cur_len <= prev_len) {
            sym = ((p - 1 - prev_at) << 16) | (prev_len - 3);
            emit = 1;
            pending = 0;
            cur_len = 2;
            p += prev_len - 1;
        } else if (pending) {
            sym = lit;
            emit = 1;
            p++;
        } else {
            pending = 1;
            p++;
        }
        lit = s0123 & 0xffu; /* the byte a literal emitted by the next iteration stands for */
        if (emit)
            SG_EMIT(sym);
    }
    FOR_GLANES
    {
        if ((uint32_t)GLANE < nstaged)
            tok[ntok - nstaged + (uint32_t)GLANE] = LV(stg);
    }
    if (sd_blk != 0xffffffffu)
        SG_TRACE_FLUSH();
    ON_GLANE0
    {
        SgWave *me = &lds->wv[s];
        me->exit_kind = exit_kind;
        me->exit_p = p;
        me->exit_len = cur_len;
        me->exit_at = cur_at;
        me->exit_pending = (uint32_t)pending;
        me->ntok = ntok;
    }
    WAVE_SYNC();
}

/* phase 2 (every wave): take segments off the queue, last segment first -- by the time
 * a segment's parser runs past its end, the segments behind it have usually been
 * parsed, so it finds a hand-over.  Segment 0 starts from the state the previous
 * super-step ended in, every other one fresh.  In a redo round (phase 3 found a parser
 * that gave up) wave 0 parses on from that parser's exact state. */
template <bool TABLE>
DEV void sg_phase_parse(const LzJob &job, SgLds *lds, const SgScratch &scr, int w)
{
    const uint32_t S0 = GUNI(lds->S0);
    const uint32_t nact = sg_nact(S0, job.n);
    const int redo = (int)GUNI(lds->redo);
    if (redo && (w != 0 || GGROUP != 0)) /* one parser carries an exact state on */
        return;
    for (int round = 0;; round++) {
        uint32_t s, sp, slen = 2, sat = 0, spend = 0;
        if (redo) {
            if (round)
                break;
            const uint32_t k = GUNI(lds->redo_from);
            s = GUNI(lds->redo_seg);
            sp = GUNI(lds->wv[k].exit_p);
            slen = GUNI(lds->wv[k].exit_len);
            sat = GUNI(lds->wv[k].exit_at);
            spend = GUNI(lds->wv[k].exit_pending);
        } else {
            LANEVAR(uint32_t, got);
            FOR_GLANES
            {
                LV(got) = 0;
                if (GLANE == 0)
                    LV(got) = LDS_FETCH_ADD_U32(&lds->queue, 0xffffffffu);
            }
            const uint32_t old = GREADLANE(got, 0);
            if (old == 0 || old > nact) /* empty (the counter may have gone below zero) */
                break;
            s = GUNI(lds->emu_ascending) ? nact - old : old - 1;
            sp = S0 + s * SG_G;
            if (s == 0) {
                sp = GUNI(lds->wv[0].start_p);
                slen = GUNI(lds->wv[0].start_len);
                sat = GUNI(lds->wv[0].start_at);
                spend = GUNI(lds->wv[0].start_pending);
            }
        }
        SG_COUNT(2, redo ? 0x10000 + s : s);
        sg_parse_segment<TABLE>(job, lds, scr, s, sp, slen, sat, (int)spend);
        SG_COUNT(3, 0);
    }
}

/* append tokens [from, to) of one wave's round to the buffer's symbol stream,
 * cutting a block whenever it holds lit_bufsize-1 symbols */
DEV void sg_append(const LzJob &job, SgOut *o, const uint32_t *tok, uint32_t from, uint32_t to,
                   int may_cut, uint32_t cut_delta, uint32_t need)
{
    uint32_t nsyms = UNI(o->nsyms), nblocks = UNI(o->nblocks);
    uint32_t blk_sym0 = UNI(o->blk_sym0), blk_in0 = UNI(o->blk_in0), cov = UNI(o->cov);
    uint32_t i = from;
    /* the next batch's tokens are asked for while this one is worked on: the resolver is one wave, the
     * others wait for it, and every batch would otherwise begin with a trip to memory */
    LANEVAR(uint32_t, pf);
    uint32_t pf_at = 0xffffffffu;
    FOR_LANES { LV(pf) = 0; }
    while (i < to) {
        /* never let a batch run across a block boundary */
        uint32_t room = may_cut ? job.cfg.sym_cap - (nsyms - blk_sym0) : WAVE;
        uint32_t cnt = to - i < WAVE ? to - i : WAVE;
        if (cnt > room)
            cnt = room;
        LANEVAR(uint32_t, tlen);
        LANEVAR(uint32_t, tex);
        LANEVAR(uint32_t, tk);
        if (pf_at == i) {
            FOR_LANES { LV(tk) = (uint32_t)LANE < cnt ? LV(pf) : 0u; }
        } else {
            FOR_LANES { LV(tk) = (uint32_t)LANE < cnt ? tok[i + (uint32_t)LANE] : 0u; }
        }
        if (i + cnt < to) {
            pf_at = i + cnt;
            FOR_LANES { LV(pf) = pf_at + (uint32_t)LANE < to ? tok[pf_at + (uint32_t)LANE] : 0u; }
        }
        FOR_LANES
        {
            const uint32_t t = LV(tk);
            LV(tlen) = (uint32_t)LANE < cnt ? ((t >> 16) ? (t & 0xffu) + 3u : 1u) : 0u;
        }
        uint32_t total;
        WAVE_EXSCAN(tlen, tex, total);
        FOR_LANES
        {
            if ((uint32_t)LANE < cnt)
                job.syms[nsyms + (uint32_t)LANE] = LV(tk);
        }
        /* start position of the batch's last token: the iteration that emitted it ran
         * cut_delta positions later (1 for the lazy parse, 0 for the greedy ones), which is
         * where the block is cut */
        const uint32_t last_start = cov + READLANE(tex, cnt - 1);
        nsyms += cnt;
        cov += total;
        i += cnt;
        if (may_cut && nsyms - blk_sym0 == job.cfg.sym_cap) {
            ON_LANE0
            {
                ZdBlockRec *b = &job.blocks[nblocks];
                b->sym_begin = blk_sym0;
                b->sym_count = job.cfg.sym_cap;
                b->in_begin = blk_in0;
                b->in_len = cov - blk_in0;
                const uint32_t base = sg_base_at(job.cfg, last_start + cut_delta, job.n, need);
                const uint64_t wend = (uint64_t)base + 2ull * job.cfg.wsize;
                b->stored_ok = blk_in0 >= base ? 1u : 0u;
                b->last = 0;
                b->cut = ZD_CUT_FULL;
                b->wend = wend < 0xffffffffull ? (uint32_t)wend : 0xffffffffu;
                b->at = last_start + cut_delta;
            }
            nblocks++;
            blk_sym0 = nsyms;
            blk_in0 = cov;
        }
    }
    ON_LANE0
    {
        o->nsyms = nsyms;
        o->nblocks = nblocks;
        o->blk_sym0 = blk_sym0;
        o->blk_in0 = blk_in0;
        o->cov = cov;
    }
    WAVE_SYNC();
}

/* phase 3 (wave 0): follow the chain of hand-overs from segment 0 and collect the
 * tokens; where a parser gave up, ask for a redo round and come back */
DEV void sg_phase_resolve(const LzJob &job, SgLds *lds, const SgScratch &scr, int w)
{
    if (w != 0)
        return;
    const uint32_t S0 = UNI(lds->S0);
    uint32_t k = UNI(lds->chain), ft = UNI(lds->chain_ft);
    for (;;) {
        const uint32_t kind = UNI(lds->wv[k].exit_kind);
        const uint32_t xp = UNI(lds->wv[k].exit_p);
        sg_append(job, &lds->out, scr.tok + k * SG_TOKCAP, ft, UNI(lds->wv[k].ntok), 1, 1u, ZD_MIN_LOOKAHEAD);
        if (kind == SG_EXIT_SYNCED) {
            const uint32_t t = (xp - S0) / SG_G;
            ft = UNI(scr.sidx[t * SG_TRACE + (xp - S0) % SG_G]);
            k = t;
            continue;
        }
        if (kind == SG_EXIT_UNSYNCED) {
            /* xp < end of the super-step, so it lies in a later segment of it */
            ON_LANE0
            {
                lds->redo = 1;
                lds->redo_from = k;
                lds->redo_seg = (xp - S0) / SG_G;
                lds->chain = (xp - S0) / SG_G;
                lds->chain_ft = 0;
            }
            WAVE_SYNC();
            return;
        }
        if (kind == SG_EXIT_LAST) {
            ON_LANE0
            {
                lds->wv[0].start_p = xp;
                lds->wv[0].start_len = lds->wv[k].exit_len;
                lds->wv[0].start_at = lds->wv[k].exit_at;
                lds->wv[0].start_pending = lds->wv[k].exit_pending;
                lds->S0 += SG_SPAN;
                lds->redo = 0;
            }
            WAVE_SYNC();
            return;
        }
        /* SG_EXIT_END: the parse reached the end of the input (src/deflate.c:2108-2117) */
        if (UNI(lds->wv[k].exit_pending)) {
            /* the last byte goes out as a literal; _tr_tally's "block full" answer is
             * ignored here (src/deflate.c:2109-2112), so no cut */
            const uint32_t c = UNI(job.in[xp - 1]);
            ON_LANE0 { lds->cstage[0] = c; }
            WAVE_SYNC();
            sg_append(job, &lds->out, lds->cstage, 0, 1, 0, 1u, ZD_MIN_LOOKAHEAD);
        }
        ON_LANE0
        {
            /* a run of sections that is not the end of its stream: Z_FULL_FLUSH, the block only
             * if it holds anything (src/deflate.c:2118-2120) */
            const uint32_t cutting = !job.more || lds->out.nsyms != lds->out.blk_sym0;
            if (cutting) {
                ZdBlockRec *b = &job.blocks[lds->out.nblocks];
                b->sym_begin = lds->out.blk_sym0;
                b->sym_count = lds->out.nsyms - lds->out.blk_sym0;
                b->in_begin = lds->out.blk_in0;
                b->in_len = job.n - lds->out.blk_in0;
                b->stored_ok = lds->out.blk_in0 >= sg_base(job.cfg, job.n, job.n) ? 1u : 0u;
                b->last = job.more ? 0u : 1u;
                b->cut = ZD_CUT_END;
                b->wend = 0xffffffffu; /* at the end of the input everything given has been read */
                b->at = job.n;
                /* a run with joints goes on from here (sg_next_phase) */
                lds->out.nblocks++;
                lds->out.blk_sym0 = lds->out.nsyms;
                lds->out.blk_in0 = job.n;
            }
            job.out->nsyms = lds->out.nsyms;
            job.out->nblocks = lds->out.nblocks;
            lds->redo = 0;
            lds->finished = 1;
        }
        WAVE_SYNC();
        return;
    }
}

/* A run with joints (zsc_dev.h ZdSched) is parsed in phases, each with the input given so far
 * as its n: a joint of kind 1 is where a phase reaches its end -- the owed literal and the cut
 * are what SG_EXIT_END does anyway -- and the next phase starts there like a super-step, with
 * the window and the output carried over.  A joint of kind 0 only changes n, and nothing the
 * parse did before it depends on n when the cut lies MIN_LOOKAHEAD or more before the old end
 * (sections.h sec_seg_ok; the others go to the wave-per-buffer parser). */
DEV void sg_next_phase(SgLds *lds, int w, uint32_t from)
{
    if (w != 0)
        return;
    ON_LANE0
    {
        lds->S0 = from;
        lds->finished = 0;
        lds->queue = 0;
        lds->redo = 0;
        lds->redo_seg = lds->redo_from = 0;
        lds->chain = lds->chain_ft = 0;
        lds->wv[0].start_p = from;
        lds->wv[0].start_len = 2;
        lds->wv[0].start_at = 0;
        lds->wv[0].start_pending = 0;
    }
    WAVE_SYNC();
}

/* the end of the phase that starts with joint *si: joints of kind 0 are folded into it */
DEV uint32_t sg_phase_end(const LzJob &job, uint32_t n_now, uint32_t *si)
{
    while (*si < job.nsched && UNI(job.sched[*si].kind) == 0u) {
        n_now = UNI(job.sched[*si].new_n); /* all of them can be folded: sections.h sec_seg_ok */
        (*si)++;
    }
    return n_now;
}

/* before the first super-step (wave 0) */
DEV void sg_init(SgLds *lds, int w)
{
    if (w != 0)
        return;
    ON_LANE0
    {
        lds->S0 = 0;
        lds->finished = 0;
        lds->queue = 0;
        lds->redo = 0;
        lds->redo_seg = lds->redo_from = 0;
        lds->chain = lds->chain_ft = 0;
        lds->emu_ascending = 0;
        lds->lo = lds->hi = lds->wrap_base = 0;
        lds->out.nsyms = lds->out.nblocks = lds->out.blk_sym0 = lds->out.blk_in0 = lds->out.cov = 0;
        lds->wv[0].start_p = 0;
        lds->wv[0].start_len = 2;
        lds->wv[0].start_at = 0;
        lds->wv[0].start_pending = 0;
    }
    WAVE_SYNC();
}

#ifdef SG_GROUP
#undef ZSC_GROUP
#define ZSC_GROUP 64
#include "wave_group.h"
#endif

#endif
