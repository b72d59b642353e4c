/*
 * bit_emit.h -- kernels 4a/4b: stream layout and bit packing.
 *
 * layout_buffer()  one thread per buffer: prefix-sums the exact block sizes that
 *                  huff_plan produced into bit offsets (a stored block re-aligns
 *                  to a byte, reference _tr_stored_block src/trees.c:838-849; the
 *                  last block pads to a byte, bi_windup :1081-1092), writes the
 *                  zlib/gzip header and trailer (reference src/deflate.c:1029-1082,
 *                  1271-1286) and zeroes the few words where neighbouring blocks
 *                  meet, so that blocks can be packed independently.
 * emit_block()     one wavefront per block: restates compress_block + send_bits
 *                  (reference src/trees.c:292-304,948-993).  Each lane turns one
 *                  symbol into <= 48 bits (code, length extra, distance code,
 *                  distance extra), a wave prefix sum of the bit counts gives every
 *                  lane its position, lanes OR their bits into a 512-byte LDS
 *                  staging area and the wave stores the completed words coalesced.
 *                  Only the first three and the last word of a block, which a
 *                  neighbour may also touch, go out as atomic ORs.
 */
#ifndef ZSC_BIT_EMIT_H
#define ZSC_BIT_EMIT_H

#include "huff_plan.h"
#include "wave.h"
#include "zsc_dev.h"

#define BE_STAGE_WORDS 128
#define BE_ZONE_WORDS 3u /* words at the start of a block that are written with atomic OR */

typedef struct {
    uint16_t lcode[288];
    uint16_t dcode[32];
    uint8_t llen[288];
    uint8_t dlen[32];
    uint32_t stage[BE_STAGE_WORDS];
} BeLds;

/* wave-uniform emitter state */
typedef struct {
    uint32_t *out;    /* the buffer's stream, as words */
    uint32_t pos;     /* next bit */
    uint32_t wbase;   /* global word index of stage[0] */
    uint32_t zone_lo; /* words below this index: atomic */
    uint32_t zone_hi; /* words at or above this index: atomic */
} BeState;

DEV void be_store_word(const BeState &st, uint32_t g, uint32_t v)
{
    if (g < st.zone_lo || g >= st.zone_hi)
        GLOBAL_OR_U32(&st.out[g], v);
    else
        st.out[g] = v;
}

/* move completed words of the staging area to memory, keep the partial one */
DEV void be_flush(BeLds *lds, BeState &st, int final)
{
    const uint32_t nfull = (st.pos >> 5) - st.wbase;
    const uint32_t nout = nfull + ((final && (st.pos & 31u)) ? 1u : 0u);
    for (uint32_t i = 0; i < nout; i += WAVE) {
        FOR_LANES
        {
            uint32_t k = i + (uint32_t)LANE;
            if (k < nout)
                be_store_word(st, st.wbase + k, lds->stage[k]);
        }
    }
    const uint32_t carry = lds->stage[nfull < BE_STAGE_WORDS ? nfull : 0];
    WAVE_SYNC();
    for (uint32_t i = 0; i <= nfull && i < BE_STAGE_WORDS; i += WAVE) {
        FOR_LANES
        {
            uint32_t k = i + (uint32_t)LANE;
            if (k <= nfull && k < BE_STAGE_WORDS)
                lds->stage[k] = 0;
        }
    }
    WAVE_SYNC();
    ON_LANE0 { lds->stage[0] = final ? 0u : carry; }
    WAVE_SYNC();
    st.wbase += nfull;
}

/* every lane contributes `nb` (0..48) bits `bits`; appended in lane order */
#define BE_STEP(lds, st, bits, nb)                                              \
    do {                                                                        \
        LANEVAR(uint32_t, _ex);                                                 \
        uint32_t _tot;                                                          \
        WAVE_EXSCAN(nb, _ex, _tot);                                             \
        FOR_LANES                                                               \
        {                                                                       \
            if (LV(nb)) {                                                       \
                uint32_t _b = (st).pos + LV(_ex) - ((st).wbase << 5);           \
                uint32_t _w = _b >> 5, _sh = _b & 31u;                          \
                uint64_t _v = LV(bits);                                         \
                LDS_OR_U32(&(lds)->stage[_w], (uint32_t)(_v << _sh));           \
                if (_sh + LV(nb) > 32u) {                                       \
                    uint64_t _r = _v >> (32u - _sh);                            \
                    LDS_OR_U32(&(lds)->stage[_w + 1], (uint32_t)_r);            \
                    if (_sh + LV(nb) > 64u)                                     \
                        LDS_OR_U32(&(lds)->stage[_w + 2], (uint32_t)(_r >> 32)); \
                }                                                               \
            }                                                                   \
        }                                                                       \
        (st).pos += _tot;                                                       \
        WAVE_SYNC();                                                            \
        be_flush((lds), (st), 0);                                               \
    } while (0)

/* byte write that respects the atomic zones (stored blocks) */
DEV void be_store_byte(const BeState &st, uint32_t byte_idx, uint32_t v)
{
    const uint32_t g = byte_idx >> 2;
    if (g < st.zone_lo || g >= st.zone_hi)
        GLOBAL_OR_U32(&st.out[g], v << (8u * (byte_idx & 3u)));
    else
        ((uint8_t *)st.out)[byte_idx] = (uint8_t)v;
}

DEV uint32_t be_block_end_bit(const ZdBlockRec *rec, const ZdBlockPlan *plan)
{
    if (plan->type == ZD_BT_STORED)
        return ((plan->bit_off + 3u + 7u) & ~7u) + 32u + 8u * rec->in_len;
    return plan->bit_off + plan->body_bits;
}

/* pack one block; `in` = the buffer's input, `syms` = the block's first symbol,
 * `out` = the buffer's stream (4-byte aligned) */
DEV void emit_block(const uint8_t *in, const uint32_t *syms, const ZdBlockRec *rec,
                    const ZdBlockPlan *plan, uint32_t *out, BeLds *lds)
{
    BeState st;
    st.out = out;
    st.pos = plan->bit_off;
    st.wbase = st.pos >> 5;
    st.zone_lo = st.wbase + BE_ZONE_WORDS;
    st.zone_hi = be_block_end_bit(rec, plan) >> 5;
    const uint32_t type = plan->type;
    if (type > ZD_BT_DYNAMIC)
        return; /* the stream did not fit: layout_buffer reported Z_BUF_ERROR */

    for (int i = 0; i < BE_STAGE_WORDS; i += WAVE) {
        FOR_LANES { lds->stage[i + LANE] = 0; }
    }
    WAVE_SYNC();

    if (type == ZD_BT_STORED) {
        /* reference _tr_stored_block, src/trees.c:838-849 */
        {
            LANEVAR(uint64_t, bits);
            LANEVAR(uint32_t, nb);
            FOR_LANES
            {
                LV(bits) = rec->last; /* BTYPE 00 */
                LV(nb) = LANE == 0 ? 3u : 0u;
            }
            BE_STEP(lds, st, bits, nb);
        }
        be_flush(lds, st, 1);
        const uint32_t at = (plan->bit_off + 3u + 7u) >> 3; /* first byte after the header */
        const uint32_t len = rec->in_len;
        FOR_LANES
        {
            if (LANE < 4) {
                uint32_t v = LANE == 0 ? len : LANE == 1 ? len >> 8 : LANE == 2 ? ~len : ~len >> 8;
                be_store_byte(st, at + (uint32_t)LANE, v & 0xffu);
            }
        }
        for (uint32_t i = 0; i < len; i += WAVE) {
            FOR_LANES
            {
                uint32_t k = i + (uint32_t)LANE;
                if (k < len)
                    be_store_byte(st, at + 4u + k, in[rec->in_begin + k]);
            }
        }
        return;
    }

    /* code tables into LDS */
    for (int i = 0; i < 288; i += WAVE) {
        FOR_LANES
        {
            int s = i + LANE;
            if (s < 288) {
                if (type == ZD_BT_STATIC) {
                    /* static_ltree, reference src/trees.c:110-169 */
                    uint32_t l = hp_static_llen((uint32_t)s);
                    uint32_t c = s < 144 ? 0x30u + (uint32_t)s
                                 : s < 256 ? 0x190u + (uint32_t)(s - 144)
                                 : s < 280 ? (uint32_t)(s - 256)
                                           : 0xC0u + (uint32_t)(s - 280);
                    lds->lcode[s] = (uint16_t)hp_bitrev(c, (int)l);
                    lds->llen[s] = (uint8_t)l;
                } else {
                    lds->lcode[s] = s < HP_LCODES ? plan->lcode[s] : (uint16_t)0;
                    lds->llen[s] = s < HP_LCODES ? plan->llen[s] : (uint8_t)0;
                }
            }
        }
    }
    FOR_LANES
    {
        if (LANE < 32) {
            if (type == ZD_BT_STATIC) {
                lds->dcode[LANE] = (uint16_t)hp_bitrev((uint32_t)LANE, 5);
                lds->dlen[LANE] = 5;
            } else {
                lds->dcode[LANE] = LANE < HP_DCODES ? plan->dcode[LANE] : (uint16_t)0;
                lds->dlen[LANE] = LANE < HP_DCODES ? plan->dlen[LANE] : (uint8_t)0;
            }
        }
    }

    WAVE_SYNC();
    /* 3-bit block header, then (dynamic) the tree description, 8 bits per lane */
    {
        LANEVAR(uint64_t, bits);
        LANEVAR(uint32_t, nb);
        FOR_LANES
        {
            LV(bits) = ((type == ZD_BT_STATIC ? 1u : 2u) << 1) | rec->last;
            LV(nb) = LANE == 0 ? 3u : 0u;
        }
        BE_STEP(lds, st, bits, nb);
    }
    if (type == ZD_BT_DYNAMIC) {
        const uint32_t hb = plan->hdr_bits;
        for (uint32_t i = 0; i < hb; i += WAVE * 8) {
            LANEVAR(uint64_t, bits);
            LANEVAR(uint32_t, nb);
            FOR_LANES
            {
                uint32_t b0 = i + 8u * (uint32_t)LANE;
                uint32_t n = b0 < hb ? (hb - b0 < 8u ? hb - b0 : 8u) : 0u;
                LV(nb) = n;
                LV(bits) = n ? (uint64_t)(plan->hdr[b0 >> 3] & ((1u << n) - 1u)) : 0ull;
            }
            BE_STEP(lds, st, bits, nb);
        }
    }

    /* the symbols, reference compress_block src/trees.c:958-990 */
    const uint32_t count = rec->sym_count;
    for (uint32_t s = 0; s < count; s += WAVE) {
        LANEVAR(uint64_t, bits);
        LANEVAR(uint32_t, nb);
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            uint64_t v = 0;
            uint32_t n = 0;
            if (i < count) {
                uint32_t sym = syms[i];
                uint32_t dist = sym >> 16, lc = sym & 0xffu;
                if (dist == 0) {
                    v = lds->lcode[lc];
                    n = lds->llen[lc];
                } else {
                    uint32_t c = hp_len_code(lc);
                    v = lds->lcode[257 + c];
                    n = lds->llen[257 + c];
                    uint32_t xb = (uint32_t)hp_extra(0, (int)(257 + c));
                    if (xb)
                        v |= (uint64_t)(lc - hp_len_base(c)) << n;
                    n += xb;
                    dist--;
                    c = hp_dist_code(dist);
                    v |= (uint64_t)lds->dcode[c] << n;
                    n += lds->dlen[c];
                    xb = (uint32_t)hp_extra(1, (int)c);
                    if (xb)
                        v |= (uint64_t)(dist - hp_dist_base(c)) << n;
                    n += xb;
                }
            }
            LV(bits) = v;
            LV(nb) = n;
        }
        BE_STEP(lds, st, bits, nb);
    }
    /* END_BLOCK */
    {
        LANEVAR(uint64_t, bits);
        LANEVAR(uint32_t, nb);
        FOR_LANES
        {
            LV(bits) = lds->lcode[256];
            LV(nb) = LANE == 0 ? lds->llen[256] : 0u;
        }
        BE_STEP(lds, st, bits, nb);
    }
    be_flush(lds, st, 1);
}

/* one thread per buffer */
DEV void layout_buffer(const ZdBuf *buf, const ZdParseOut *po, const ZdBlockRec *recs,
                       ZdBlockPlan *plans, ZdResult *res, uint8_t *out)
{
    if (po->nblocks > buf->max_blocks) {
        res->status = -2; /* the parser gave up (cannot happen unless there is a bug): fail loudly */
        res->out_len = 0;
        return;
    }
    const uint32_t hdr_bytes = buf->wrap == 1 ? 2u : buf->wrap == 2 ? 10u : 0u;
    const uint32_t trl_bytes = buf->wrap == 1 ? 4u : buf->wrap == 2 ? 8u : 0u;
    uint32_t bit = hdr_bytes * 8u;
    for (uint32_t i = 0; i < po->nblocks; i++) {
        plans[i].bit_off = bit;
        bit = be_block_end_bit(&recs[i], &plans[i]);
        if (recs[i].last)
            bit = (bit + 7u) & ~7u; /* bi_windup */
    }
    res->bits = bit - hdr_bytes * 8u;
    const uint32_t body_end = (bit + 7u) >> 3; /* a whole number of bytes unless the run goes on (ZdBuf.more) */
    const uint32_t total = body_end + trl_bytes;
    res->out_len = total;
    if (total > buf->out_cap) {
        res->status = -5; /* Z_BUF_ERROR */
        for (uint32_t i = 0; i < po->nblocks; i++)
            plans[i].type = 0xffu; /* nothing is emitted */
        return;
    }
    res->status = 0;
    uint32_t *w = (uint32_t *)out;
    const uint32_t cap_words = (buf->out_cap + 3u) >> 2;
    for (uint32_t i = 0; i < po->nblocks; i++) {
        const uint32_t first = plans[i].bit_off >> 5;
        for (uint32_t k = 0; k < BE_ZONE_WORDS; k++)
            if (first + k < cap_words)
                w[first + k] = 0;
        const uint32_t endw = be_block_end_bit(&recs[i], &plans[i]) >> 5;
        if (endw < cap_words)
            w[endw] = 0;
        if (endw + 1 < cap_words)
            w[endw + 1] = 0; /* byte padding after the last block can spill one word */
    }
    if (buf->wrap == 1) {
        /* reference src/deflate.c:1031-1049 */
        uint32_t h = (8u + ((buf->wbits - 8u) << 4)) << 8;
        const uint32_t lvl = buf->level;
        const uint32_t lf = (buf->strategy >= 2 || lvl < 2) ? 0u : lvl < 6 ? 1u : lvl == 6 ? 2u : 3u;
        h |= lf << 6;
        h += 31u - h % 31u;
        out[0] = (uint8_t)(h >> 8);
        out[1] = (uint8_t)h;
        const uint32_t a = res->adler;
        out[body_end + 0] = (uint8_t)(a >> 24);
        out[body_end + 1] = (uint8_t)(a >> 16);
        out[body_end + 2] = (uint8_t)(a >> 8);
        out[body_end + 3] = (uint8_t)a;
    } else if (buf->wrap == 2) {
        /* reference src/deflate.c:1068-1082,1272-1281 */
        out[0] = 31;
        out[1] = 139;
        out[2] = 8;
        out[3] = out[4] = out[5] = out[6] = out[7] = 0;
        out[8] = buf->level == 9 ? 2 : (buf->strategy >= 2 || buf->level < 2) ? 4 : 0;
        out[9] = 3; /* OS_CODE unix */
        const uint32_t c = res->adler, n = buf->in_len;
        for (uint32_t k = 0; k < 4; k++) {
            out[body_end + k] = (uint8_t)(c >> (8 * k));
            out[body_end + 4 + k] = (uint8_t)(n >> (8 * k));
        }
    }
}

#endif
