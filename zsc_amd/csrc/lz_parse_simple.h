/*
 * lz_parse_simple.h -- kernel 2 for the two strategies that need no hash chains.
 *
 * Z_HUFFMAN_ONLY (reference deflate_huff, src/deflate.c:2210-2247): every byte is a
 * literal.  Z_RLE (deflate_rle, :2129-2204): the only matches are runs, distance 1.
 * The strategy is looked at before the level (src/deflate.c:1216-1219), so these
 * take over at every level 1-9.
 *
 * One wavefront per buffer, straight from HBM: there is no window to keep (a run only
 * looks one byte back).  Huffman-only is 64 positions per step.  RLE looks at 256
 * positions per step: each lane compares four bytes with their predecessors, a ballot
 * finds the first place where three equal pairs follow one another (= the previous byte
 * repeats three times, :2153-2155), everything before it goes out as literals in one
 * piece, and the run is measured 256 bytes per step.
 *
 * Blocks are cut every 16 383 symbols like _tr_tally does; whether a block may be stored
 * depends on where the window base stands at that moment, and the base moves when
 * fill_window is called -- at a lookahead of 258 for deflate_rle (:2141), of 0 for
 * deflate_huff (:2218).
 */
#ifndef ZSC_LZ_PARSE_SIMPLE_H
#define ZSC_LZ_PARSE_SIMPLE_H

#include "lz_parse_seg.h"

#define SP_TOK 256u

typedef struct {
    uint32_t tok[SP_TOK];
    SgOut out;
} SpLds;

DEV void sp_begin(SpLds *lds)
{
    ON_LANE0 { lds->out.nsyms = lds->out.nblocks = lds->out.blk_sym0 = lds->out.blk_in0 = lds->out.cov = 0; }
    WAVE_SYNC();
}

/* the last block (FLUSH_BLOCK(s, 1), :2197-2198 / :2240-2241) */
DEV void sp_finish(const LzJob &job, SpLds *lds, uint32_t need)
{
    ON_LANE0
    {
        const uint32_t cutting = !job.more || lds->out.nsyms != lds->out.blk_sym0; /* as in sg_phase_resolve */
        if (cutting) {
            ZdBlockRec *b = &job.blocks[lds->out.nblocks];
            b->sym_begin = lds->out.blk_sym0;
            b->sym_count = lds->out.nsyms - lds->out.blk_sym0;
            b->in_begin = lds->out.blk_in0;
            b->in_len = job.n - lds->out.blk_in0;
            b->stored_ok = lds->out.blk_in0 >= sg_base_at(job.cfg, job.n, job.n, need) ? 1u : 0u;
            b->last = job.more ? 0u : 1u;
            b->cut = ZD_CUT_END;
            b->wend = 0xffffffffu;
            b->at = job.n;
        }
        job.out->nsyms = lds->out.nsyms;
        job.out->nblocks = lds->out.nblocks + cutting;
    }
    WAVE_SYNC();
}

DEV void lz_parse_huff(const LzJob &job, SpLds *lds)
{
    sp_begin(lds);
    for (uint32_t p = 0; p < job.n; p += SP_TOK) {
        const uint32_t cnt = job.n - p < SP_TOK ? job.n - p : SP_TOK;
        FOR_LANES
        {
            for (uint32_t j = (uint32_t)LANE; j < cnt; j += WAVE)
                lds->tok[j] = job.in[p + j];
        }
        WAVE_SYNC();
        sg_append(job, &lds->out, lds->tok, 0, cnt, 1, 0u, 1u);
    }
    sp_finish(job, lds, 1u);
}

DEV void lz_parse_rle(const LzJob &job, SpLds *lds)
{
    sp_begin(lds);
    const uint32_t n = job.n;
    uint32_t p = 0;
    while (p < n) {
        /* lane l looks at bytes p+4l-1 .. p+4l+5: pairs (i-1, i) for its own four positions
         * and the two after them */
        LANEVAR(uint32_t, starts); /* bit k: a run of three equal pairs starts at p+4l+k */
        LANEVAR(int, any);
        FOR_LANES
        {
            const uint32_t x = p + 4u * (uint32_t)LANE;
            uint8_t b[7];
            for (uint32_t k = 0; k < 7u; k++) {
                const uint32_t i = x + k; /* byte index + 1 */
                b[k] = (i >= 1u && i - 1u < n) ? job.in[i - 1u] : (uint8_t)0;
            }
            uint32_t e = 0; /* bit k: byte x+k equals its predecessor (and both exist) */
            for (uint32_t k = 0; k < 6u; k++) {
                const uint32_t i = x + k;
                if (i >= 1u && i < n && b[k] == b[k + 1u])
                    e |= 1u << k;
            }
            /* a match at i needs i > 0 (:2150) and three bytes of lookahead (:2150) */
            uint32_t sbits = e & (e >> 1) & (e >> 2) & 0xfu;
            for (uint32_t k = 0; k < 4u; k++)
                if (x + k + 3u > n)
                    sbits &= ~(1u << k);
            if (x >= p + SP_TOK)
                sbits = 0;
            LV(starts) = sbits;
            LV(any) = sbits != 0;
        }
        const uint64_t am = BALLOT(any);
        uint32_t f = p + SP_TOK; /* first position of this step that starts a run */
        if (am != 0) {
            const int l0 = CTZ64(am);
            f = p + 4u * (uint32_t)l0 + (uint32_t)CTZ32(READLANE(starts, l0));
        }
        if (f > n)
            f = n;
        /* literals up to there */
        if (f > p) {
            const uint32_t cnt = f - p;
            FOR_LANES
            {
                for (uint32_t j = (uint32_t)LANE; j < cnt; j += WAVE)
                    lds->tok[j] = job.in[p + j];
            }
            WAVE_SYNC();
            sg_append(job, &lds->out, lds->tok, 0, cnt, 1, 0u, ZD_MAX_MATCH + 1u);
            p = f;
        }
        if (am == 0 || p >= n)
            continue;
        /* the run at p: bytes equal to in[p-1], at most 258 and at most what is left (:2156-2169) */
        const uint32_t prev = UNI(job.in[p - 1u]);
        const uint32_t cap = n - p < ZD_MAX_MATCH ? n - p : ZD_MAX_MATCH;
        uint32_t len = cap;
        for (uint32_t k0 = 0; k0 < cap; k0 += WAVE) {
            LANEVAR(int, differs);
            FOR_LANES
            {
                const uint32_t k = k0 + (uint32_t)LANE;
                LV(differs) = k < cap && job.in[p + k] != prev;
            }
            const uint64_t dm = BALLOT(differs);
            if (dm != 0) {
                len = k0 + (uint32_t)CTZ64(dm);
                break;
            }
        }
        ON_LANE0 { lds->tok[0] = (1u << 16) | (len - 3u); }
        WAVE_SYNC();
        sg_append(job, &lds->out, lds->tok, 0, 1, 1, 0u, ZD_MAX_MATCH + 1u);
        p += len;
    }
    sp_finish(job, lds, ZD_MAX_MATCH + 1u);
}

/* Both, symbol by symbol, for a run of sections with joints (zsc_dev.h ZdSched): how much input
 * deflate() has been given changes on the way, and with it the window's slides and the
 * lookahead that bounds a run.  Rare (an output slice of max_block_len has to run out while a
 * block is flushed, with one of these strategies), so plain: one symbol per step, lane 0
 * writes it. */
DEV void lz_parse_simple_joints(const LzJob &job, SpLds *lds)
{
    (void)lds;
    const int huff = job.strategy == 2u;
    const uint32_t need = huff ? 1u : ZD_MAX_MATCH + 1u; /* :2218 / :2141 */
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
    uint32_t p = 0;
    for (;;) {
        st.it = p;
        uint32_t look = st.data_end - p;
        if (look < need) {
            lz_refill(job, st, p);
            look = st.data_end - p;
            if (look == 0) {
                if (!lz_joint_at_end(job, st, p))
                    break;
                continue;
            }
        }
        uint32_t len = 0;
        if (!huff && look >= 3u && p > 0u) {
            const uint32_t prev = UNI(job.in[p - 1u]);
            if (UNI(job.in[p]) == prev && UNI(job.in[p + 1u]) == prev && UNI(job.in[p + 2u]) == prev) {
                const uint32_t cap = look < ZD_MAX_MATCH ? look : ZD_MAX_MATCH;
                len = cap;
                for (uint32_t k0 = 0; k0 < cap; k0 += WAVE) {
                    LANEVAR(int, differs);
                    FOR_LANES
                    {
                        const uint32_t k = k0 + (uint32_t)LANE;
                        LV(differs) = k < cap && job.in[p + k] != prev;
                    }
                    const uint64_t dm = BALLOT(differs);
                    if (dm != 0) {
                        len = k0 + (uint32_t)CTZ64(dm);
                        break;
                    }
                }
            }
        }
        const uint32_t tok = len >= 3u ? (1u << 16) | (len - 3u) : UNI(job.in[p]);
        ON_LANE0 { job.syms[st.nsyms] = tok; }
        st.nsyms++;
        p += len >= 3u ? len : 1u;
        if (st.nsyms - st.blk_sym0 == job.cfg.sym_cap)
            (void)lz_cut(job, st, p, 0, ZD_CUT_FULL);
    }
    lz_cut_end(job, st, p);
    ON_LANE0
    {
        job.out->nsyms = st.nsyms;
        job.out->nblocks = st.nblocks;
    }
    WAVE_SYNC();
}

#endif
