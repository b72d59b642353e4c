/*
 * lz_parse_seg.h -- kernel 2 for long buffers: several wavefronts parse ONE buffer.
 *
 * The lazy parse (reference deflate_slow, src/deflate.c:1989-2122) is serial, and its
 * 32 KiB window has to sit in LDS, so a wave-per-buffer kernel gets four waves onto a
 * CU and is bound by single-wave latency.  This kernel shares one window between
 * SG_W waves of a workgroup and lets every wave parse its own 1 KiB segment of the
 * same buffer at the same time -- speculatively, because the state the serial parse
 * would arrive in at the start of a segment is not known yet:
 *
 *   - hash chains do not depend on the parse at levels 4-9 (every position is
 *     inserted, src/deflate.c:2018,2069-2075), and the window base is a function
 *     of the position alone, so a parser that stands at position x with NO match
 *     pending ("fresh": match_available == 0, match_length == 2) continues
 *     identically whatever happened before x;
 *   - wave k starts fresh at its segment start and records every position it
 *     visits fresh (a bitmap in LDS, plus its token count there);
 *   - wave k-1, the only one whose start is known to be right, keeps parsing past
 *     its segment end until it is fresh at a position wave k also visited fresh:
 *     from there on wave k's tokens ARE the serial parse's tokens.  Text resyncs
 *     within a few tokens;
 *   - if no common position turns up within SG_OV bytes (a long run, say), wave k-1
 *     stops with its exact state and wave k parses its segment again from that
 *     state -- the super-step degrades towards the serial parse, never to a
 *     different result.
 *
 * After each super-step (SG_W segments) wave 0 walks the chain of hand-overs and
 * appends the valid token ranges to the buffer's symbol stream, cutting blocks
 * every 16 383 symbols exactly as _tr_tally does (include/zsc/deflate.h:338-354).
 * The symbol stream, block records and therefore every later kernel and the final
 * bytes are identical to the wave-per-buffer parser's.
 */
#ifndef ZSC_LZ_PARSE_SEG_H
#define ZSC_LZ_PARSE_SEG_H

#include "lz_parse.h"

#define SG_W 8                      /* waves per workgroup = segments per super-step */
#define SG_G 1024u                  /* positions per segment */
#define SG_OV 512u                  /* how far past its segment a wave looks for a hand-over */
#define SG_SPAN (SG_W * SG_G)
#define SG_TRACE (SG_G + SG_OV)     /* positions a wave records */
#define SG_TOKCAP (SG_G + SG_OV + 320u) /* tokens one wave can emit in one round */

#define SG_EXIT_SYNCED 1u
#define SG_EXIT_UNSYNCED 2u
#define SG_EXIT_LAST 3u
#define SG_EXIT_END 4u

typedef struct {
    uint32_t run;                                   /* parse in the coming round */
    uint32_t start_p, start_len, start_at, start_pending;
    uint32_t exit_kind, exit_p, exit_len, exit_at, exit_pending;
    uint32_t ntok;                                  /* tokens emitted in its last round */
    uint32_t first_tok;                             /* first token that belongs to the serial parse */
} SgWave;

struct SgLds {
    static constexpr uint32_t RING = 45056u, CHUNK = 2048u;
    static constexpr bool HAS_INS = false;
    uint8_t ring[RING + 512];
    uint32_t stage[SG_W][WAVE];
    uint32_t trace[SG_W][SG_TRACE / 32];
    SgWave wv[SG_W];
    /* workgroup state */
    uint32_t S0;                  /* first position of the current super-step */
    uint32_t again, finished, chain;
    uint32_t lo, hi, wrap_base;   /* window ring */
    /* the buffer's symbol stream being assembled */
    uint32_t nsyms, nblocks, blk_sym0, blk_in0, cov;
    uint32_t cstage[WAVE];
};

/* scratch in HBM per workgroup */
typedef struct {
    uint32_t *tok;  /* SG_W * SG_TOKCAP tokens */
    uint16_t *sidx; /* SG_W * SG_TRACE: tokens emitted before a fresh position */
} SgScratch;

/* the window base the serial parse has at a loop top at position p: every slide of
 * fill_window (src/deflate.c:1563-1570) whose condition holds at p has happened */
DEV uint32_t sg_base(uint32_t p, uint32_t n)
{
    uint32_t base = 0;
    for (;;) {
        uint64_t end = (uint64_t)base + 2ull * ZD_TILE;
        uint32_t data_end = end < n ? (uint32_t)end : n;
        if ((uint64_t)p + ZD_MIN_LOOKAHEAD > data_end && p - base >= ZD_TILE + ZD_MAX_DIST)
            base += ZD_TILE;
        else
            return base;
    }
}

DEV void sg_flush_stage(uint32_t *tok, SgLds *lds, int w, uint32_t ntok, uint32_t nstaged)
{
    const uint32_t first = ntok - nstaged;
    FOR_LANES
    {
        if ((uint32_t)LANE < nstaged)
            tok[first + (uint32_t)LANE] = lds->stage[w][LANE];
    }
}

/* phase 1 (wave 0): slide the window to the new super-step, decide who parses */
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
    const uint32_t want = want64 < job.n ? (uint32_t)want64 : job.n;
    while (st.hi < want)
        lz_load_chunk<L>(job, lds, st);
    ON_LANE0
    {
        lds->lo = st.lo;
        lds->hi = st.hi;
        lds->wrap_base = st.wrap_base;
        for (int k = 1; k < SG_W; k++) {
            const uint32_t a = S0 + (uint32_t)k * SG_G;
            lds->wv[k].run = a < job.n ? 1u : 0u;
            lds->wv[k].start_p = a;
            lds->wv[k].start_len = 2;
            lds->wv[k].start_at = 0;
            lds->wv[k].start_pending = 0;
            lds->wv[k].first_tok = 0;
            lds->wv[k].ntok = 0;
        }
        lds->wv[0].run = 1; /* its start state was carried over */
        lds->wv[0].first_tok = 0;
        lds->chain = 0;
        lds->again = 0;
    }
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)LANE; i < SG_W * (SG_TRACE / 32); i += WAVE)
            (&lds->trace[0][0])[i] = 0;
    }
    WAVE_SYNC();
}

/* phase 2 (every wave that has `run` set): parse one segment */
DEV void sg_phase_parse(const LzJob &job, SgLds *lds, const SgScratch &scr, int w)
{
    typedef SgLds L;
    if (!UNI(lds->wv[w].run))
        return;
    LzState st;
    st.lo = UNI(lds->lo);
    st.hi = UNI(lds->hi);
    st.wrap_base = UNI(lds->wrap_base);
    st.nsyms = st.nstaged = st.nblocks = st.blk_sym0 = st.blk_in0 = st.pr_hi = 0;

    const uint32_t S0 = UNI(lds->S0);
    const uint64_t E64 = (uint64_t)S0 + SG_SPAN;
    const uint32_t E = E64 < job.n ? (uint32_t)E64 : job.n; /* end of the super-step */
    const uint32_t a_w = S0 + (uint32_t)w * SG_G;
    const uint32_t e_w = a_w + SG_G < E ? a_w + SG_G : E;
    const int last_seg = e_w == E;

    uint32_t p = UNI(lds->wv[w].start_p);
    uint32_t cur_len = UNI(lds->wv[w].start_len), cur_at = UNI(lds->wv[w].start_at);
    int pending = (int)UNI(lds->wv[w].start_pending);
    st.base = sg_base(p, job.n);
    {
        uint64_t end = (uint64_t)st.base + 2ull * ZD_TILE;
        st.data_end = end < job.n ? (uint32_t)end : job.n;
    }
    uint32_t *tok = scr.tok + (uint32_t)w * SG_TOKCAP;
    uint16_t *sidx = scr.sidx + (uint32_t)w * SG_TRACE;
    uint32_t ntok = 0, nstaged = 0, exit_kind = 0;
    /* rank[] / hib[] of 64 consecutive positions, one per lane: a search reads them with
     * v_readlane instead of a dependent global load */
    LANEVAR(uint32_t, rkhb);
    uint32_t rk_at = 0;
    int rk_valid = 0;
    FOR_LANES { LV(rkhb) = 0; }

    for (;;) {
        uint32_t look = st.data_end - p;
        if (look < ZD_MIN_LOOKAHEAD) {
            lz_refill(job, st, p);
            look = st.data_end - p;
            if (look == 0) {
                exit_kind = SG_EXIT_END;
                break;
            }
        }
        const int fresh = !pending && cur_len == 2;
        if (p >= e_w) {
            if (last_seg) {
                exit_kind = SG_EXIT_LAST;
                break;
            }
            if (fresh && p - (a_w + SG_G) < SG_TRACE) {
                const uint32_t r = p - (a_w + SG_G); /* index in the successor's trace */
                if ((UNI(lds->trace[w + 1][r >> 5]) >> (r & 31u)) & 1u) {
                    exit_kind = SG_EXIT_SYNCED;
                    break;
                }
            }
            if (p >= e_w + SG_OV) {
                exit_kind = SG_EXIT_UNSYNCED;
                break;
            }
        }
        if (fresh && p >= a_w && p - a_w < SG_TRACE) {
            const uint32_t r = p - a_w;
            ON_LANE0
            {
                sidx[r] = (uint16_t)ntok;
                lds->trace[w][r >> 5] |= 1u << (r & 31u);
            }
            WAVE_SYNC();
        }

        const uint32_t prev_len = cur_len, prev_at = cur_at;
        cur_len = 2;
        if (look >= 3 && prev_len < job.cfg.lazy) {
            LzSearch sc;
            sc.p = p;
            const uint32_t w0 = UNI(ld_u32(&lds->ring[lz_ridx<L>(st, p)]));
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
            if (!rk_valid || p - rk_at >= WAVE) {
                rk_valid = 1;
                rk_at = p;
                FOR_LANES
                {
                    const uint32_t x = p + (uint32_t)LANE;
                    LV(rkhb) = x + 2 < job.n ? ((uint32_t)job.rank[x] | ((uint32_t)job.hib[x] << 16)) : 0u;
                }
            }
            const uint32_t rh = READLANE(rkhb, p - rk_at);
            const int32_t hiA = (int32_t)(rh & 0xffffu) - 1;
            const int32_t hiB = tile ? (int32_t)(int16_t)(rh >> 16) : -1;
            int verdict = 0;
            LZ_WALK_RUN(runA, hiA + WAVE, tile << 15, LZ_MEMB_ALL, verdict);
            if (verdict == 1 && tile != 0) {
                verdict = 0;
                LZ_WALK_RUN(runA - ZD_TILE, hiB + WAVE, (tile - 1) << 15, LZ_MEMB_ALL, verdict);
            }
            if (verdict != 3 && sc.head_seen) {
                cur_at = sc.where;
                cur_len = sc.best < look ? sc.best : look;
                if (cur_len <= 5 &&
                    (job.strategy == 1 || (cur_len == 3 && p - cur_at > ZD_TOO_FAR)))
                    cur_len = 2;
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
            sym = UNI(lds->ring[lz_ridx<L>(st, p - 1)]);
            emit = 1;
            p++;
        } else {
            pending = 1;
            p++;
        }
        if (emit) {
            ON_LANE0 { lds->stage[w][nstaged] = sym; }
            WAVE_SYNC();
            nstaged++;
            ntok++;
            if (nstaged == WAVE) {
                sg_flush_stage(tok, lds, w, ntok, nstaged);
                nstaged = 0;
            }
        }
    }
    if (nstaged)
        sg_flush_stage(tok, lds, w, ntok, nstaged);
    ON_LANE0
    {
        SgWave *me = &lds->wv[w];
        me->exit_kind = exit_kind;
        me->exit_p = p;
        me->exit_len = cur_len;
        me->exit_at = cur_at;
        me->exit_pending = (uint32_t)pending;
        me->ntok = ntok;
        me->run = 0;
    }
    WAVE_SYNC();
}

/* append tokens [from, to) of one wave's round to the buffer's symbol stream,
 * cutting a block whenever it holds ZD_SYM_CAP symbols */
DEV void sg_append(const LzJob &job, SgLds *lds, const uint32_t *tok, uint32_t from, uint32_t to,
                   int may_cut)
{
    uint32_t nsyms = UNI(lds->nsyms), nblocks = UNI(lds->nblocks);
    uint32_t blk_sym0 = UNI(lds->blk_sym0), blk_in0 = UNI(lds->blk_in0), cov = UNI(lds->cov);
    uint32_t i = from;
    while (i < to) {
        /* never let a batch run across a block boundary */
        uint32_t room = may_cut ? ZD_SYM_CAP - (nsyms - blk_sym0) : WAVE;
        uint32_t cnt = to - i < WAVE ? to - i : WAVE;
        if (cnt > room)
            cnt = room;
        LANEVAR(uint32_t, tlen);
        LANEVAR(uint32_t, tex);
        LANEVAR(uint32_t, tk);
        FOR_LANES
        {
            uint32_t t = (uint32_t)LANE < cnt ? tok[i + (uint32_t)LANE] : 0u;
            LV(tk) = t;
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
         * one position later (lazy parse), which is where the block is cut */
        const uint32_t last_start = cov + READLANE(tex, cnt - 1);
        nsyms += cnt;
        cov += total;
        i += cnt;
        if (may_cut && nsyms - blk_sym0 == ZD_SYM_CAP) {
            ON_LANE0
            {
                ZdBlockRec *b = &job.blocks[nblocks];
                b->sym_begin = blk_sym0;
                b->sym_count = ZD_SYM_CAP;
                b->in_begin = blk_in0;
                b->in_len = cov - blk_in0;
                b->stored_ok = blk_in0 >= sg_base(last_start + 1, job.n) ? 1u : 0u;
                b->last = 0;
            }
            nblocks++;
            blk_sym0 = nsyms;
            blk_in0 = cov;
        }
    }
    ON_LANE0
    {
        lds->nsyms = nsyms;
        lds->nblocks = nblocks;
        lds->blk_sym0 = blk_sym0;
        lds->blk_in0 = blk_in0;
        lds->cov = cov;
    }
    WAVE_SYNC();
}

/* phase 3 (wave 0): follow the chain of hand-overs, collect tokens, schedule redos */
DEV void sg_phase_resolve(const LzJob &job, SgLds *lds, const SgScratch &scr, int w)
{
    if (w != 0)
        return;
    uint32_t k = UNI(lds->chain);
    for (;;) {
        const uint32_t kind = UNI(lds->wv[k].exit_kind);
        const uint32_t xp = UNI(lds->wv[k].exit_p);
        sg_append(job, lds, scr.tok + k * SG_TOKCAP, UNI(lds->wv[k].first_tok), UNI(lds->wv[k].ntok), 1);
        if (kind == SG_EXIT_SYNCED) {
            const uint32_t j = k + 1;
            const uint32_t a_j = UNI(lds->S0) + j * SG_G;
            const uint32_t ft = UNI(scr.sidx[j * SG_TRACE + (xp - a_j)]);
            ON_LANE0 { lds->wv[j].first_tok = ft; }
            WAVE_SYNC();
            k = j;
            continue;
        }
        if (kind == SG_EXIT_UNSYNCED) {
            /* the next segment is parsed again, this time from the true state */
            const uint32_t j = k + 1;
            ON_LANE0
            {
                lds->wv[j].run = 1;
                lds->wv[j].start_p = xp;
                lds->wv[j].start_len = lds->wv[k].exit_len;
                lds->wv[j].start_at = lds->wv[k].exit_at;
                lds->wv[j].start_pending = lds->wv[k].exit_pending;
                lds->wv[j].first_tok = 0;
                lds->chain = j;
                lds->again = 1;
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
                lds->again = 0;
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
            sg_append(job, lds, lds->cstage, 0, 1, 0);
        }
        ON_LANE0
        {
            ZdBlockRec *b = &job.blocks[lds->nblocks];
            b->sym_begin = lds->blk_sym0;
            b->sym_count = lds->nsyms - lds->blk_sym0;
            b->in_begin = lds->blk_in0;
            b->in_len = job.n - lds->blk_in0;
            b->stored_ok = lds->blk_in0 >= sg_base(job.n, job.n) ? 1u : 0u;
            b->last = 1;
            job.out->nsyms = lds->nsyms;
            job.out->nblocks = lds->nblocks + 1;
            lds->again = 0;
            lds->finished = 1;
        }
        WAVE_SYNC();
        return;
    }
}

/* before the first super-step (wave 0) */
DEV void sg_init(SgLds *lds, int w)
{
    if (w != 0)
        return;
    ON_LANE0
    {
        lds->S0 = 0;
        lds->again = 0;
        lds->finished = 0;
        lds->chain = 0;
        lds->lo = lds->hi = lds->wrap_base = 0;
        lds->nsyms = lds->nblocks = lds->blk_sym0 = lds->blk_in0 = lds->cov = 0;
        lds->wv[0].start_p = 0;
        lds->wv[0].start_len = 2;
        lds->wv[0].start_at = 0;
        lds->wv[0].start_pending = 0;
    }
    WAVE_SYNC();
}

#endif
