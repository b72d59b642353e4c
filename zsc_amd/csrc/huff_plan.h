/*
 * huff_plan.h -- kernel 3: per-block Huffman planning, one wavefront per block.
 *
 * Restates _tr_flush_block's decision half (reference src/trees.c:874-934):
 * symbol statistics, build_tree x3 (:559-652) with zlib's exact heap order
 * (:377-414 -- ties between equal frequency, equal depth nodes are resolved by
 * heap mechanics, so the heap is simulated, not replaced by a sort), gen_bitlen's
 * overflow repair (:477-507), scan_tree/build_bl_tree (:658-806), and the
 * stored / static / dynamic choice (:902-934).
 *
 * The 64 lanes build the symbol histogram with LDS atomics and copy tables;
 * the heap itself is a ~10k step serial algorithm run by lane 0 out of LDS
 * (6 KiB per wave, so 24+ blocks are in flight per CU to hide its latency).
 * Output: code tables ready for the bit packer, the exact size of the block in
 * bits (opt_len/static_len are exact), and the dynamic block's header bits.
 */
#ifndef ZSC_HUFF_PLAN_H
#define ZSC_HUFF_PLAN_H

#include "wave.h"
#include "zsc_dev.h"

#define HP_LCODES 286
#define HP_DCODES 30
#define HP_BLCODES 19
#define HP_HEAP (2 * HP_LCODES + 1)

typedef struct {
    uint32_t freq[HP_HEAP];
    uint16_t parent[HP_HEAP];
    uint16_t len[HP_HEAP];
    uint16_t code[HP_HEAP];
    int max_code;
} HpTree;

typedef struct {
    uint32_t freq[2 * HP_DCODES + 1];
    uint16_t parent[2 * HP_DCODES + 1];
    uint16_t len[2 * HP_DCODES + 2];
    uint16_t code[2 * HP_DCODES + 1];
    int max_code;
} HpTreeD;

typedef struct {
    uint32_t freq[2 * HP_BLCODES + 1];
    uint16_t parent[2 * HP_BLCODES + 1];
    uint16_t len[2 * HP_BLCODES + 1];
    uint16_t code[2 * HP_BLCODES + 1];
    int max_code;
} HpTreeB;

typedef struct {
    HpTree lt;
    HpTreeD dt;
    HpTreeB bt;
    uint16_t heap[HP_HEAP];
    uint8_t depth[HP_HEAP];
    uint16_t per_len[16];
    uint32_t opt_bits, static_bits;
    int heap_n, heap_top;
    /* header bit writer */
    uint8_t hdr[ZD_HDR_BYTES];
    uint32_t hdr_acc, hdr_fill, hdr_pos;
} HpLds;

/* a view of one of the three trees plus its static description
 * (reference static_tree_desc, src/trees.c:237-252) */
typedef struct {
    uint32_t *freq;
    uint16_t *parent, *len, *code;
    int *max_code;
    int elems, max_len, extra_from, has_static, is_dist;
    int kind; /* 0 lit/len, 1 dist, 2 bit-length */
} HpView;

/* extra bits: reference src/trees.c:87-94, as arithmetic */
DEV int hp_extra(int kind, int sym)
{
    if (kind == 0) {
        if (sym < 265 || sym == 285)
            return 0;
        return (sym - 261) >> 2;
    }
    if (kind == 1)
        return sym < 4 ? 0 : (sym >> 1) - 1;
    return sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
}

DEV uint32_t hp_static_llen(uint32_t c) /* reference src/trees.c:110-169 */
{
    return c < 144 ? 8u : c < 256 ? 9u : c < 280 ? 7u : 8u;
}

/* length code of len-3 and distance code of dist-1: reference src/trees.c:180-223 */
DEV uint32_t hp_len_code(uint32_t lc)
{
    if (lc < 8)
        return lc;
    if (lc == 255)
        return 28;
    uint32_t e = 29u - (uint32_t)__builtin_clz(lc); /* floor(log2 lc) - 2 */
    return 4u * e + 4u + ((lc >> e) & 3u);
}

DEV uint32_t hp_dist_code(uint32_t d)
{
    if (d < 4)
        return d;
    uint32_t e = 30u - (uint32_t)__builtin_clz(d); /* floor(log2 d) - 1 */
    return 2u * e + 2u + ((d >> e) & 1u);
}

DEV uint32_t hp_len_base(uint32_t c) /* base_length, reference src/trees.c:225-228 */
{
    if (c < 8)
        return c;
    if (c == 28)
        return 0;
    uint32_t e = (c - 4) >> 2;
    return (4u + ((c - 4) & 3u)) << e;
}

DEV uint32_t hp_dist_base(uint32_t c) /* base_dist, reference src/trees.c:230-234 */
{
    if (c < 4)
        return c;
    uint32_t e = (c >> 1) - 1;
    return (2u + (c & 1u)) << e;
}

DEV uint32_t hp_bitrev(uint32_t v, int bits) /* bi_reverse, reference src/trees.c:1046-1058 */
{
    uint32_t r = 0;
    while (bits-- > 0) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

/* smaller(), reference src/trees.c:377-379 */
DEV int hp_before(const HpView &t, const uint8_t *depth, int a, int b)
{
    return t.freq[a] < t.freq[b] || (t.freq[a] == t.freq[b] && depth[a] <= depth[b]);
}

/* pqdownheap, reference src/trees.c:387-414 */
DEV void hp_sift(HpLds *h, const HpView &t, int k)
{
    int v = h->heap[k];
    for (int j = k << 1; j <= h->heap_n; j <<= 1) {
        if (j < h->heap_n && hp_before(t, h->depth, h->heap[j + 1], h->heap[j]))
            j++;
        if (hp_before(t, h->depth, v, h->heap[j]))
            break;
        h->heap[k] = h->heap[j];
        k = j;
    }
    h->heap[k] = (uint16_t)v;
}

/* build_tree + gen_bitlen + gen_codes, reference src/trees.c:426-652 (serial, lane 0) */
DEV void hp_build(HpLds *h, const HpView &t)
{
    int top = -1;
    h->heap_n = 0;
    h->heap_top = HP_HEAP;
    for (int n = 0; n < t.elems; n++) {
        if (t.freq[n]) {
            h->heap[++h->heap_n] = (uint16_t)(top = n);
            h->depth[n] = 0;
        } else {
            t.len[n] = 0;
        }
    }
    while (h->heap_n < 2) { /* :595-610 */
        int node = top < 2 ? ++top : 0;
        h->heap[++h->heap_n] = (uint16_t)node;
        t.freq[node] = 1;
        h->depth[node] = 0;
        h->opt_bits--;
        if (t.has_static)
            h->static_bits -= t.is_dist ? 5u : hp_static_llen((uint32_t)node);
    }
    *t.max_code = top;
    for (int n = h->heap_n / 2; n >= 1; n--)
        hp_sift(h, t, n);

    int node = t.elems;
    do { /* :624-640 */
        int a = h->heap[1];
        h->heap[1] = h->heap[h->heap_n--];
        hp_sift(h, t, 1);
        int b = h->heap[1];
        h->heap[--h->heap_top] = (uint16_t)a;
        h->heap[--h->heap_top] = (uint16_t)b;
        t.freq[node] = t.freq[a] + t.freq[b];
        h->depth[node] = (uint8_t)((h->depth[a] >= h->depth[b] ? h->depth[a] : h->depth[b]) + 1);
        t.parent[a] = t.parent[b] = (uint16_t)node;
        h->heap[1] = (uint16_t)node++;
        hp_sift(h, t, 1);
    } while (h->heap_n >= 2);
    h->heap[--h->heap_top] = h->heap[1];

    /* gen_bitlen, :445-508 */
    int over = 0;
    for (int b = 0; b < 16; b++)
        h->per_len[b] = 0;
    t.len[h->heap[h->heap_top]] = 0;
    int i;
    for (i = h->heap_top + 1; i < HP_HEAP; i++) {
        int n = h->heap[i];
        int bits = t.len[t.parent[n]] + 1;
        if (bits > t.max_len) {
            bits = t.max_len;
            over++; /* internal nodes count too: :457 runs before :461 */
        }
        t.len[n] = (uint16_t)bits;
        if (n > top)
            continue;
        h->per_len[bits]++;
        int xb = n >= t.extra_from ? hp_extra(t.kind, n) : 0;
        h->opt_bits += t.freq[n] * (uint32_t)(bits + xb);
        if (t.has_static)
            h->static_bits += t.freq[n] * ((t.is_dist ? 5u : hp_static_llen((uint32_t)n)) + (uint32_t)xb);
    }
    if (over > 0) {
        do { /* :477-489 */
            int bits = t.max_len - 1;
            while (h->per_len[bits] == 0)
                bits--;
            h->per_len[bits]--;
            h->per_len[bits + 1] += 2;
            h->per_len[t.max_len]--;
            over -= 2;
        } while (over > 0);
        for (int bits = t.max_len; bits != 0; bits--) { /* :496-507 */
            int n = h->per_len[bits];
            while (n != 0) {
                int m = h->heap[--i];
                if (m > top)
                    continue;
                if (t.len[m] != (uint16_t)bits) {
                    h->opt_bits += ((uint32_t)bits - t.len[m]) * t.freq[m];
                    t.len[m] = (uint16_t)bits;
                }
                n--;
            }
        }
    }
    /* gen_codes, :518-549 */
    uint16_t next[16];
    uint32_t code = 0;
    for (int b = 1; b <= 15; b++) {
        code = (code + h->per_len[b - 1]) << 1;
        next[b] = (uint16_t)code;
    }
    for (int n = 0; n <= top; n++) {
        int l = t.len[n];
        if (l)
            t.code[n] = (uint16_t)hp_bitrev(next[l]++, l);
    }
}

/* scan_tree, reference src/trees.c:658-707 */
DEV void hp_scan(HpLds *h, uint16_t *len, int max_code)
{
    int prev = -1, next = len[0], count = 0, hi = 7, lo = 4;
    if (next == 0) {
        hi = 138;
        lo = 3;
    }
    len[max_code + 1] = 0xffff; /* guard, :674 */
    for (int n = 0; n <= max_code; n++) {
        int cur = next;
        next = len[n + 1];
        if (++count < hi && cur == next)
            continue;
        if (count < lo)
            h->bt.freq[cur] += (uint32_t)count;
        else if (cur != 0) {
            if (cur != prev)
                h->bt.freq[cur]++;
            h->bt.freq[16]++;
        } else if (count <= 10)
            h->bt.freq[17]++;
        else
            h->bt.freq[18]++;
        count = 0;
        prev = cur;
        if (next == 0) {
            hi = 138;
            lo = 3;
        } else if (cur == next) {
            hi = 6;
            lo = 3;
        } else {
            hi = 7;
            lo = 4;
        }
    }
}

/* header bit writer (send_bits into the LDS staging copy of the block header) */
DEV void hp_put(HpLds *h, uint32_t value, int nbits)
{
    h->hdr_acc |= value << h->hdr_fill;
    h->hdr_fill += (uint32_t)nbits;
    while (h->hdr_fill >= 8) {
        h->hdr[h->hdr_pos++] = (uint8_t)h->hdr_acc;
        h->hdr_acc >>= 8;
        h->hdr_fill -= 8;
    }
}

/* send_tree, reference src/trees.c:713-773 */
DEV void hp_send_lengths(HpLds *h, const uint16_t *len, int max_code)
{
    int prev = -1, next = len[0], count = 0, hi = 7, lo = 4;
    if (next == 0) {
        hi = 138;
        lo = 3;
    }
    for (int n = 0; n <= max_code; n++) {
        int cur = next;
        next = len[n + 1];
        if (++count < hi && cur == next)
            continue;
        if (count < lo) {
            do
                hp_put(h, h->bt.code[cur], h->bt.len[cur]);
            while (--count != 0);
        } else if (cur != 0) {
            if (cur != prev) {
                hp_put(h, h->bt.code[cur], h->bt.len[cur]);
                count--;
            }
            hp_put(h, h->bt.code[16], h->bt.len[16]);
            hp_put(h, (uint32_t)(count - 3), 2);
        } else if (count <= 10) {
            hp_put(h, h->bt.code[17], h->bt.len[17]);
            hp_put(h, (uint32_t)(count - 3), 3);
        } else {
            hp_put(h, h->bt.code[18], h->bt.len[18]);
            hp_put(h, (uint32_t)(count - 11), 7);
        }
        count = 0;
        prev = cur;
        if (next == 0) {
            hi = 138;
            lo = 3;
        } else if (cur == next) {
            hi = 6;
            lo = 3;
        } else {
            hi = 7;
            lo = 4;
        }
    }
}

/* plan one block: `syms` points at the block's first symbol */
DEV void huff_plan_block(const uint32_t *syms, const ZdBlockRec *rec, uint32_t strategy,
                         ZdBlockPlan *plan, HpLds *h)
{
    const uint32_t count = rec->sym_count;
    /* init_block, reference src/trees.c:336-356 */
    for (int i = 0; i < HP_HEAP; i += WAVE) {
        FOR_LANES
        {
            if (i + LANE < HP_HEAP)
                h->lt.freq[i + LANE] = 0;
        }
    }
    FOR_LANES
    {
        if (LANE < 2 * HP_DCODES + 1)
            h->dt.freq[LANE] = 0;
        if (LANE < 2 * HP_BLCODES + 1)
            h->bt.freq[LANE] = 0;
    }
    /* symbol statistics, reference include/zsc/deflate.h:338-354 */
    for (uint32_t s = 0; s < count; s += WAVE) {
        FOR_LANES
        {
            uint32_t i = s + (uint32_t)LANE;
            if (i < count) {
                uint32_t v = syms[i];
                uint32_t dist = v >> 16, lc = v & 0xff;
                if (dist == 0) {
                    LDS_ADD_U32(&h->lt.freq[lc], 1u);
                } else {
                    LDS_ADD_U32(&h->lt.freq[257 + hp_len_code(lc)], 1u);
                    LDS_ADD_U32(&h->dt.freq[hp_dist_code(dist - 1)], 1u);
                }
            }
        }
    }
    WAVE_SYNC();
    int last_bl = 0;
    uint32_t type = ZD_BT_DYNAMIC;
    ON_LANE0
    {
        h->lt.freq[256] = 1; /* END_BLOCK */
        h->opt_bits = h->static_bits = 0;
        HpView vl = {h->lt.freq, h->lt.parent, h->lt.len, h->lt.code, &h->lt.max_code,
                     HP_LCODES, 15, 257, 1, 0, 0};
        HpView vd = {h->dt.freq, h->dt.parent, h->dt.len, h->dt.code, &h->dt.max_code,
                     HP_DCODES, 15, 0, 1, 1, 1};
        HpView vb = {h->bt.freq, h->bt.parent, h->bt.len, h->bt.code, &h->bt.max_code,
                     HP_BLCODES, 7, 0, 0, 0, 2};
        hp_build(h, vl);
        hp_build(h, vd);
        /* build_bl_tree, reference src/trees.c:779-806 */
        hp_scan(h, h->lt.len, h->lt.max_code);
        hp_scan(h, h->dt.len, h->dt.max_code);
        hp_build(h, vb);
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (last_bl = HP_BLCODES - 1; last_bl >= 3; last_bl--)
            if (h->bt.len[order[last_bl]] != 0)
                break;
        h->opt_bits += 3u * ((uint32_t)last_bl + 1u) + 5u + 5u + 4u;

        /* reference src/trees.c:902-934 */
        uint32_t opt_bytes = (h->opt_bits + 3 + 7) >> 3;
        const uint32_t static_bytes = (h->static_bits + 3 + 7) >> 3;
        if (static_bytes <= opt_bytes)
            opt_bytes = static_bytes;
        if (rec->in_len + 4 <= opt_bytes && rec->stored_ok)
            type = ZD_BT_STORED;
        else if (strategy == 4 || static_bytes == opt_bytes)
            type = ZD_BT_STATIC;
        else
            type = ZD_BT_DYNAMIC;

        h->hdr_acc = h->hdr_fill = h->hdr_pos = 0;
        if (type == ZD_BT_DYNAMIC) {
            /* send_all_trees, reference src/trees.c:813-833 */
            hp_put(h, (uint32_t)(h->lt.max_code + 1 - 257), 5);
            hp_put(h, (uint32_t)(h->dt.max_code + 1 - 1), 5);
            hp_put(h, (uint32_t)(last_bl + 1 - 4), 4);
            for (int r = 0; r <= last_bl; r++)
                hp_put(h, h->bt.len[order[r]], 3);
            hp_send_lengths(h, h->lt.len, h->lt.max_code);
            hp_send_lengths(h, h->dt.len, h->dt.max_code);
        }
        plan->type = type;
        plan->hdr_bits = h->hdr_pos * 8 + h->hdr_fill;
        if (h->hdr_fill)
            h->hdr[h->hdr_pos] = (uint8_t)h->hdr_acc;
        plan->body_bits = type == ZD_BT_DYNAMIC ? 3u + h->opt_bits
                          : type == ZD_BT_STATIC ? 3u + h->static_bits : 0u;
        plan->bit_off = 0;
    }
    WAVE_SYNC();
    /* publish the tables the bit packer needs */
    for (int i = 0; i < HP_LCODES; i += WAVE) {
        FOR_LANES
        {
            int s = i + LANE;
            if (s < HP_LCODES) {
                int used = s <= h->lt.max_code && h->lt.len[s] != 0;
                plan->lcode[s] = used ? h->lt.code[s] : (uint16_t)0;
                plan->llen[s] = used ? (uint8_t)h->lt.len[s] : (uint8_t)0;
            }
        }
    }
    FOR_LANES
    {
        if (LANE < HP_DCODES) {
            int used = LANE <= h->dt.max_code && h->dt.len[LANE] != 0;
            plan->dcode[LANE] = used ? h->dt.code[LANE] : (uint16_t)0;
            plan->dlen[LANE] = used ? (uint8_t)h->dt.len[LANE] : (uint8_t)0;
        }
    }
    for (int i = 0; i < (int)ZD_HDR_BYTES; i += WAVE) {
        FOR_LANES
        {
            if (i + LANE < (int)ZD_HDR_BYTES)
                plan->hdr[i + LANE] = h->hdr[i + LANE];
        }
    }
}

#endif
