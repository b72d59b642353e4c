/*
 * zsc_oracle.c -- CPU restatement of the zsc DEFLATE hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see zsc_oracle.h.  Parity status: PINNED against
 * the compiled reference (oracle/_ref) by tests/test_oracle_vs_ref.py and the
 * vectors in tests/golden/.
 *
 * This is a restatement, not a copy: positions are absolute offsets into the
 * caller's buffer (the reference slides a 64 KiB window and rebases its hash
 * chains; here the window base B is a number and a candidate is dead when it
 * is <= B), the three stages are separate functions with explicit data
 * between them, and trees live in plain arrays.  Every rule that decides an
 * output bit cites the reference line it restates (paths relative to
 * /root/reference).
 */
#include "zsc_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* checksums                                                                 */
/* ------------------------------------------------------------------------- */

/* src/adler32.c:40-41,56-131: a = 1 + sum(d), b = sum(a), both mod 65521, the
 * modulo deferred over runs of at most 5552 bytes (largest n with
 * 255n(n+1)/2 + (n+1)(65520) < 2^32). */
uint32_t zo_adler32(uint32_t adler, const uint8_t *buf, uint32_t len)
{
    uint32_t a = adler & 0xffffu, b = (adler >> 16) & 0xffffu;
    if (buf == NULL)
        return 1u;
    while (len > 0) {
        uint32_t run = len < 5552u ? len : 5552u;
        len -= run;
        while (run--) {
            a += *buf++;
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}

/* src/crc32.c:502-528: reflected CRC-32, polynomial 0xEDB88320, pre/post
 * inverted.  The reference ships slice-by-4 tables (:58-492); any table
 * organisation yields the same value, we use slice-by-8 built at load. */
static uint32_t zo_crc_tab[8][256];

__attribute__((constructor)) static void zo_crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        zo_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            zo_crc_tab[t][i] = (zo_crc_tab[t - 1][i] >> 8) ^ zo_crc_tab[0][zo_crc_tab[t - 1][i] & 0xff];
}

uint32_t zo_crc32(uint32_t crc, const uint8_t *buf, uint32_t len)
{
    if (buf == NULL)
        return 0u;
    uint32_t c = ~crc;
    while (len && ((uintptr_t)buf & 7u)) {
        c = zo_crc_tab[0][(c ^ *buf++) & 0xff] ^ (c >> 8);
        len--;
    }
    while (len >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, buf, 4);
        memcpy(&hi, buf + 4, 4);
        lo ^= c;
        c = zo_crc_tab[7][lo & 0xff] ^ zo_crc_tab[6][(lo >> 8) & 0xff] ^
            zo_crc_tab[5][(lo >> 16) & 0xff] ^ zo_crc_tab[4][lo >> 24] ^
            zo_crc_tab[3][hi & 0xff] ^ zo_crc_tab[2][(hi >> 8) & 0xff] ^
            zo_crc_tab[1][(hi >> 16) & 0xff] ^ zo_crc_tab[0][hi >> 24];
        buf += 8;
        len -= 8;
    }
    while (len--)
        c = zo_crc_tab[0][(c ^ *buf++) & 0xff] ^ (c >> 8);
    return ~c;
}

/* ------------------------------------------------------------------------- */
/* sizing helpers                                                            */
/* ------------------------------------------------------------------------- */

/* split the public window_bits encoding: returns wrap (0 raw, 1 zlib, 2 gzip) */
static int zo_split_wbits(int window_bits, int *abs_bits)
{
    if (window_bits < 0) {
        *abs_bits = -window_bits;
        return 0;
    }
    if (window_bits > 15) {
        *abs_bits = window_bits - 16;
        return 2;
    }
    *abs_bits = window_bits;
    return 1;
}

/* src/deflate.c:761-849 with gz_head == NULL */
int zo_deflate_bound(uint32_t source_len, int level, int window_bits, int mem_level,
                     uint32_t *size_out)
{
    int wb, wrap = zo_split_wbits(window_bits, &wb);
    *size_out = 0xffffffffu;
    if (mem_level < 1 || mem_level > 9 || wb < 8 || wb > 15 || (wb == 8 && wrap != 1))
        return ZO_STREAM_ERROR;
    uint32_t wraplen = wrap == 0 ? 0u : wrap == 1 ? 10u : 18u;
    if (wb != 15 || mem_level != 8 || level == 0)
        *size_out = source_len + ((source_len + 7) >> 3) + ((source_len + 63) >> 6) + 5 + wraplen;
    else
        *size_out = source_len + (source_len >> 12) + (source_len >> 14) + (source_len >> 25) +
                    13 - 6 + wraplen;
    return ZO_OK;
}

/* src/zsc_compress.c:207-236: bound, plus 4 bytes per possible section, bounded again */
int zo_compress_max_output(uint32_t source_len, uint32_t max_block_len, int level,
                           int window_bits, int mem_level, uint32_t *size_out)
{
    uint32_t first;
    int err = zo_deflate_bound(source_len, level, window_bits, mem_level, &first);
    if (err != ZO_OK)
        return err;
    uint32_t sections = first / max_block_len + 1;
    return zo_deflate_bound(source_len + sections * 4u, level, window_bits, mem_level, size_out);
}

/* src/deflate.c:857-902 */
int zo_compress_work_size(int window_bits, int mem_level, uint32_t state_size, uint32_t *size_out)
{
    int wb;
    (void)zo_split_wbits(window_bits, &wb);
    *size_out = 0xffffffffu;
    if (wb == 8)
        wb = 9;
    if (mem_level < 1 || mem_level > 9 || wb < 8 || wb > 15)
        return ZO_STREAM_ERROR;
    uint32_t w = 1u << wb;
    *size_out = state_size + w * 2u + w * 2u * 2u + (1u << (mem_level + 7)) * 2u +
                (1u << (mem_level + 6)) * 4u;
    return ZO_OK;
}

/* src/inflate.c:249-276 */
int zo_uncompress_work_size(int window_bits, uint32_t state_size, uint32_t *size_out)
{
    int wb = window_bits;
    if (wb < 0)
        wb = -wb;
    else if (wb < 48)
        wb &= 15;
    if (wb && (wb < 8 || wb > 15))
        return ZO_STREAM_ERROR;
    *size_out = state_size + (1u << wb);
    return ZO_OK;
}

/* ------------------------------------------------------------------------- */
/* stage P: LZ77 parse                                                       */
/* ------------------------------------------------------------------------- */

/* src/deflate.c:146-158 */
typedef struct {
    uint16_t good, lazy, nice, chain;
    uint8_t slow;
} zo_level_cfg;

static const zo_level_cfg zo_levels[10] = {
    {0, 0, 0, 0, 0},        {4, 4, 8, 4, 0},        {4, 5, 16, 8, 0},      {4, 6, 32, 32, 0},
    {4, 4, 16, 16, 1},      {8, 16, 32, 32, 1},     {8, 16, 128, 128, 1},  {8, 32, 128, 256, 1},
    {32, 128, 258, 1024, 1}, {32, 258, 258, 4096, 1}};

#define ZO_MIN_LOOKAHEAD 262u /* MAX_MATCH + MIN_MATCH + 1, include/zsc/deflate.h:303 */
#define ZO_TOO_FAR 4096u      /* src/deflate.c:130 */

typedef struct {
    const uint8_t *in;
    uint32_t n;
    /* geometry */
    uint32_t wsize, wmask, max_dist; /* max_dist = wsize - 262, deflate.h:308 */
    uint32_t hmask, hshift;
    uint32_t sym_cap; /* lit_bufsize - 1: a block is cut when it holds this many symbols */
    zo_level_cfg cfg;
    int strategy;
    /* chains: absolute positions; a value <= base is the reference's NIL */
    uint32_t *head, *link;
    uint32_t base;     /* absolute offset of window[0] (multiple of wsize) */
    uint32_t data_end; /* first byte not yet in the window */
    /* output */
    zo_symbol *syms;
    uint32_t nsyms;
    zo_block *blocks;
    uint32_t nblocks;
    uint32_t block_begin_sym, block_begin_in;
    /* where the parse stands (the parse functions can be left at an event and re-entered) */
    uint32_t p;            /* strstart */
    uint32_t cur_len, cur_at; /* match_length, match_start */
    int have_pending;      /* match_available */
    uint32_t glen, gat;    /* deflate_fast's match_length / match_start */
    uint32_t insert;       /* strings at the end of the data seen so far that could not be indexed yet */
} zo_parser;

enum { ZO_EV_BLOCK = 1, ZO_EV_END = 2 }; /* a block was cut / the data seen so far is used up */

/* UPDATE_HASH applied to three consecutive bytes, src/deflate.c:174-175,1593-1596 */
static inline uint32_t zo_hash3(const zo_parser *z, uint32_t p)
{
    uint32_t h = z->in[p];
    h = ((h << z->hshift) ^ z->in[p + 1]) & z->hmask;
    h = ((h << z->hshift) ^ z->in[p + 2]) & z->hmask;
    return h;
}

/* INSERT_STRING, src/deflate.c:186-189: returns the previous chain head */
static inline uint32_t zo_insert(zo_parser *z, uint32_t p)
{
    uint32_t h = zo_hash3(z, p);
    uint32_t old = z->head[h];
    z->link[p & z->wmask] = old;
    z->head[h] = p;
    return old;
}

/* fill_window, src/deflate.c:1532-1614, for a caller that supplied all input up
 * front: when the parse position is wsize+max_dist or more past the window base
 * the window slides by wsize (:1563-1570), then input is read to the end of the
 * 2*wsize window (:1589). */
static void zo_refill(zo_parser *z, uint32_t p)
{
    if (p - z->base >= z->wsize + z->max_dist)
        z->base += z->wsize;
    uint64_t end = (uint64_t)z->base + 2u * (uint64_t)z->wsize;
    z->data_end = end < z->n ? (uint32_t)end : z->n;
    /* :1591-1612: strings that ended the previous call's input without three bytes to hash */
    uint32_t look = z->data_end - p;
    if (z->insert && look + z->insert >= 3) {
        uint32_t str = p - z->insert;
        while (z->insert) {
            (void)zo_insert(z, str);
            str++;
            z->insert--;
            if (look + z->insert < 3)
                break;
        }
    }
}

/* longest_match, src/deflate.c:1400-1518.  `cur` is the chain head (already
 * known to be alive and within max_dist); returns the match length and sets
 * *where when a strictly longer match than prev_len was found.
 *
 * zsc-specific rule (:1462-1469 vs :1508-1512): a candidate that fails the four
 * byte pre-check advances the chain WITHOUT consuming chain budget; only
 * candidates that get the full comparison do.
 *
 * The reference compares up to 258 bytes even past the end of the data and
 * clamps the result to the lookahead (:1514-1517).  Here comparisons stop at the
 * lookahead; the outcome is the same because a comparison that reaches the
 * lookahead is >= nice_match (:1436-1438) and ends the search either way, and a
 * search entered with prev_len >= lookahead can only return the lookahead. */
static uint32_t zo_longest(const zo_parser *z, uint32_t p, uint32_t cur, uint32_t prev_len,
                           uint32_t *where)
{
    const uint8_t *in = z->in;
    uint32_t look = z->data_end - p;
    uint32_t budget = z->cfg.chain;
    uint32_t best = prev_len;
    uint32_t nice = z->cfg.nice;
    uint32_t floor_pos = (p - z->base > z->max_dist) ? p - z->max_dist : z->base;
    uint32_t cap = look < 258u ? look : 258u;

    if (prev_len >= z->cfg.good)
        budget >>= 2;
    if (nice > look)
        nice = look;
    if (best >= look)
        return look;

    for (;;) {
        const uint8_t *m = in + cur, *s = in + p;
        if (m[best] == s[best] && m[best - 1] == s[best - 1] && m[0] == s[0] && m[1] == s[1]) {
            uint32_t len = 2; /* byte 2 follows from equal hash + equal bytes 0,1 (:1473-1480) */
            while (len < cap && m[len] == s[len])
                len++;
            if (len > best) {
                *where = cur;
                best = len;
                if (len >= nice)
                    break;
            }
            budget--;
        }
        cur = z->link[cur & z->wmask];
        if (cur <= floor_pos || budget == 0)
            break;
    }
    return best < look ? best : look;
}

/* FLUSH_BLOCK_ONLY, src/deflate.c:1660-1668: close the current block at input
 * position `upto`.  stored_ok restates `block_start >= 0` (:1661). */
static void zo_cut_block(zo_parser *z, uint32_t upto, int last)
{
    zo_block *b = &z->blocks[z->nblocks++];
    b->sym_begin = z->block_begin_sym;
    b->sym_count = z->nsyms - z->block_begin_sym;
    b->in_begin = z->block_begin_in;
    b->in_len = upto - z->block_begin_in;
    b->stored_ok = (uint8_t)(z->block_begin_in >= z->base);
    b->last = (uint8_t)last;
    b->pad[0] = b->pad[1] = 0;
    z->block_begin_sym = z->nsyms;
    z->block_begin_in = upto;
}

/* _tr_tally_lit / _tr_tally_dist, include/zsc/deflate.h:338-354: returns the
 * "block is full" flag */
static inline int zo_put_literal(zo_parser *z, uint8_t c)
{
    zo_symbol *s = &z->syms[z->nsyms++];
    s->dist = 0;
    s->lc = c;
    s->pad = 0;
    return z->nsyms - z->block_begin_sym == z->sym_cap;
}

static inline int zo_put_match(zo_parser *z, uint32_t dist, uint32_t len)
{
    zo_symbol *s = &z->syms[z->nsyms++];
    s->dist = (uint16_t)dist;
    s->lc = (uint8_t)(len - 3u);
    s->pad = 0;
    return z->nsyms - z->block_begin_sym == z->sym_cap;
}

/* deflate_slow, src/deflate.c:1989-2122 (levels 4-9): runs until a block is cut or the data
 * seen so far (z->n) is used up */
static int zo_run_lazy(zo_parser *z)
{
    for (;;) {
        uint32_t p = z->p;
        uint32_t look = z->data_end - p;
        if (look < ZO_MIN_LOOKAHEAD) {
            zo_refill(z, p);
            look = z->data_end - p;
            if (look == 0)
                return ZO_EV_END;
        }
        uint32_t head = 0; /* absolute 0 is never > base, i.e. NIL */
        if (look >= 3)
            head = zo_insert(z, p);

        uint32_t prev_len = z->cur_len, prev_at = z->cur_at;
        z->cur_len = 2;
        if (head > z->base && prev_len < z->cfg.lazy && p - head <= z->max_dist) {
            z->cur_len = zo_longest(z, p, head, prev_len, &z->cur_at);
            /* :2038-2047 */
            if (z->cur_len <= 5 && (z->strategy == 1 || (z->cur_len == 3 && p - z->cur_at > ZO_TOO_FAR)))
                z->cur_len = 2;
        }
        if (prev_len >= 3 && z->cur_len <= prev_len) {
            /* the match found one byte back wins (:2052-2082) */
            uint32_t last_insert = p + look - 3;
            int full = zo_put_match(z, p - 1 - prev_at, prev_len);
            for (uint32_t k = prev_len - 2; k != 0; k--) {
                p++;
                if (p <= last_insert)
                    (void)zo_insert(z, p);
            }
            z->have_pending = 0;
            z->cur_len = 2;
            p++;
            z->p = p;
            if (full) {
                zo_cut_block(z, p, 0);
                return ZO_EV_BLOCK;
            }
        } else if (z->have_pending) {
            /* :2084-2097: previous byte goes out as a literal */
            int full = zo_put_literal(z, z->in[p - 1]);
            z->p = p + 1;
            if (full) {
                zo_cut_block(z, p, 0); /* FLUSH_BLOCK_ONLY comes before strstart++ (:2091-2095) */
                return ZO_EV_BLOCK;
            }
        } else {
            z->have_pending = 1;
            z->p = p + 1;
        }
    }
}

/* deflate_fast, src/deflate.c:1886-1982 (levels 1-3) */
static int zo_run_greedy(zo_parser *z)
{
    for (;;) {
        uint32_t p = z->p;
        uint32_t look = z->data_end - p;
        if (look < ZO_MIN_LOOKAHEAD) {
            zo_refill(z, p);
            look = z->data_end - p;
            if (look == 0)
                return ZO_EV_END;
        }
        uint32_t head = 0;
        if (look >= 3)
            head = zo_insert(z, p);
        if (head > z->base && p - head <= z->max_dist)
            z->glen = zo_longest(z, p, head, 2, &z->gat); /* prev_length stays MIN_MATCH-1 */

        int full;
        uint32_t len = z->glen;
        if (len >= 3) {
            full = zo_put_match(z, p - z->gat, len);
            look -= len;
            if (len <= z->cfg.lazy /* max_insert_length */ && look >= 3) {
                /* :1940-1950: short match, index every covered position */
                for (len--; len != 0; len--) {
                    p++;
                    (void)zo_insert(z, p);
                }
                p++;
                z->glen = 0;
            } else {
                p += len; /* :1951-1962: long match, skip without indexing */
                z->glen = 0;
            }
        } else {
            full = zo_put_literal(z, z->in[p]);
            p++;
        }
        z->p = p;
        if (full) {
            zo_cut_block(z, p, 0);
            return ZO_EV_BLOCK;
        }
    }
}

/* deflate_rle, src/deflate.c:2129-2204 (Z_RLE): only runs -- matches at distance 1 */
static int zo_run_rle(zo_parser *z)
{
    for (;;) {
        uint32_t p = z->p;
        uint32_t look = z->data_end - p;
        if (look <= 258) { /* :2141 */
            zo_refill(z, p);
            look = z->data_end - p;
            if (look == 0)
                return ZO_EV_END;
        }
        uint32_t len = 0;
        if (look >= 3 && p > 0) { /* strstart > 0: only the very first byte has window index 0 */
            uint8_t prev = z->in[p - 1];
            if (z->in[p] == prev && z->in[p + 1] == prev && z->in[p + 2] == prev) {
                /* the reference scans on into whatever lies behind the data and clamps to
                 * the lookahead afterwards (:2166-2169): the run inside the data, at most 258 */
                len = 3;
                while (len < 258 && len < look && z->in[p + len] == prev)
                    len++;
            }
        }
        int full;
        if (len >= 3) {
            full = zo_put_match(z, 1, len);
            p += len;
        } else {
            full = zo_put_literal(z, z->in[p]);
            p++;
        }
        z->p = p;
        if (full) {
            zo_cut_block(z, p, 0);
            return ZO_EV_BLOCK;
        }
    }
}

/* deflate_huff, src/deflate.c:2210-2247 (Z_HUFFMAN_ONLY): every byte a literal */
static int zo_run_huff(zo_parser *z)
{
    for (;;) {
        uint32_t p = z->p;
        if (z->data_end - p == 0) { /* :2218 */
            zo_refill(z, p);
            if (z->data_end - p == 0)
                return ZO_EV_END;
        }
        int full = zo_put_literal(z, z->in[p]);
        z->p = p + 1;
        if (full) {
            zo_cut_block(z, p + 1, 0);
            return ZO_EV_BLOCK;
        }
    }
}

/* the parse function deflate() picks, src/deflate.c:1216-1219 */
static int zo_run(zo_parser *z)
{
    if (z->strategy == 2)
        return zo_run_huff(z);
    if (z->strategy == 3)
        return zo_run_rle(z);
    return z->cfg.slow ? zo_run_lazy(z) : zo_run_greedy(z);
}

/* what the parse functions do when the input is used up and a flush was asked for, before
 * they flush the block: the literal still owed (:2108-2112), and the strings at the very end
 * that the next call's fill_window has to index (:2113, :1975; none for rle/huff :2195,2238) */
static void zo_end_of_input(zo_parser *z)
{
    if (z->strategy != 2 && z->strategy != 3) {
        if (z->cfg.slow && z->have_pending) {
            (void)zo_put_literal(z, z->in[z->p - 1]);
            z->have_pending = 0;
        }
        z->insert = z->p < 2 ? z->p : 2;
    } else {
        z->insert = 0;
    }
}

/* a parser at the start of a stream, or after a full flush (CLEAR_HASH, strstart = 0,
 * src/deflate.c:1244-1250); the symbol and block arrays carry on */
static void zo_parser_restart(zo_parser *z, const uint8_t *in)
{
    z->in = in;
    z->n = 0;
    z->base = 0;
    z->data_end = 0;
    z->p = 0;
    z->cur_len = 2;
    z->cur_at = 0;
    z->have_pending = 0;
    z->glen = z->gat = 0;
    z->insert = 0;
    z->block_begin_in = 0;
    memset(z->head, 0, ((size_t)z->hmask + 1) * sizeof(uint32_t));
}

int zo_parse(const uint8_t *in, uint32_t n, int level, int wbits, int mem_level, int strategy,
             zo_symbol *syms, uint32_t *nsyms, zo_block *blocks, uint32_t *nblocks)
{
    if (level < 1 || level > 9 || wbits < 9 || wbits > 15 || mem_level < 1 || mem_level > 9)
        return ZO_STREAM_ERROR;
    zo_parser z;
    memset(&z, 0, sizeof z);
    z.in = in;
    z.n = n;
    z.wsize = 1u << wbits;
    z.wmask = z.wsize - 1;
    z.max_dist = z.wsize - ZO_MIN_LOOKAHEAD;
    uint32_t hbits = (uint32_t)mem_level + 7u;
    z.hmask = (1u << hbits) - 1;
    z.hshift = (hbits + 2u) / 3u; /* src/deflate.c:350 */
    z.sym_cap = (1u << (mem_level + 6)) - 1u;
    z.cfg = zo_levels[level];
    z.strategy = strategy;
    z.head = (uint32_t *)calloc((size_t)z.hmask + 1, sizeof(uint32_t));
    z.link = (uint32_t *)calloc(z.wsize, sizeof(uint32_t));
    if (!z.head || !z.link) {
        free(z.head);
        free(z.link);
        return ZO_MEM_ERROR;
    }
    z.syms = syms;
    z.blocks = blocks;
    z.cur_len = 2;
    while (zo_run(&z) != ZO_EV_END) {
    }
    zo_end_of_input(&z);
    zo_cut_block(&z, z.p, 1); /* FLUSH_BLOCK(s, 1), :2114-2117 */
    *nsyms = z.nsyms;
    *nblocks = z.nblocks;
    free(z.head);
    free(z.link);
    return ZO_OK;
}

/* ------------------------------------------------------------------------- */
/* stage H: Huffman code construction                                        */
/* ------------------------------------------------------------------------- */

#define ZO_LCODES 286
#define ZO_DCODES 30
#define ZO_BLCODES 19
#define ZO_HEAP (2 * ZO_LCODES + 1)

/* src/trees.c:87-97,209-234 */
static const uint8_t zo_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                         2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint8_t zo_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,  4,  5,  5,  6,
                                          6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t zo_bl_extra[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
static const uint8_t zo_bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
static const uint16_t zo_len_base[29] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,  12,  14,  16,  20, 24,
                                         28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 0};
static const uint16_t zo_dist_base[30] = {0,   1,   2,   3,   4,   6,    8,    12,   16,   24,
                                          32,  48,  64,  96,  128, 192,  256,  384,  512,  768,
                                          1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576};

/* length code of (len-3): src/trees.c:209-223 as arithmetic */
static inline uint32_t zo_len_code(uint32_t lc)
{
    if (lc < 8)
        return lc;
    if (lc == 255)
        return 28;
    uint32_t e = 0, v = lc;
    while (v >= 8) { /* highest set bit position - 2 */
        v >>= 1;
        e++;
    }
    return 4 * e + 4 + ((lc >> e) & 3);
}

/* distance code of (dist-1): src/trees.c:180-207, deflate.h:326-327 as arithmetic */
static inline uint32_t zo_dist_code(uint32_t d)
{
    if (d < 4)
        return d;
    uint32_t e = 0, v = d;
    while (v >= 4) {
        v >>= 1;
        e++;
    }
    return 2 * e + 2 + ((d >> e) & 1);
}

static inline uint32_t zo_static_llen(uint32_t c) /* src/trees.c:110-169 */
{
    return c < 144 ? 8 : c < 256 ? 9 : c < 280 ? 7 : 8;
}

static uint32_t zo_bitrev(uint32_t v, int bits) /* src/trees.c:1046-1058 */
{
    uint32_t r = 0;
    while (bits-- > 0) {
        r = (r << 1) | (v & 1);
        v >>= 1;
    }
    return r;
}

typedef struct {
    uint16_t freq[ZO_HEAP]; /* leaves then internal nodes */
    uint16_t parent[ZO_HEAP];
    uint16_t len[ZO_HEAP];
    uint16_t code[ZO_HEAP];
    int max_code;
} zo_tree;

typedef struct {
    int elems;
    int max_len;
    int extra_from;       /* first symbol that has extra bits */
    const uint8_t *extra; /* extra bit counts, indexed sym - extra_from */
    int has_static;       /* accumulate static_len */
    int is_dist;
} zo_tree_kind;

static const zo_tree_kind zo_kind_l = {ZO_LCODES, 15, 257, zo_len_extra, 1, 0};
static const zo_tree_kind zo_kind_d = {ZO_DCODES, 15, 0, zo_dist_extra, 1, 1};
static const zo_tree_kind zo_kind_bl = {ZO_BLCODES, 7, 0, zo_bl_extra, 0, 0};

typedef struct {
    zo_tree lt, dt, bt;
    uint32_t opt_bits, static_bits; /* opt_len / static_len, src/trees.c:354 */
    int heap[ZO_HEAP];
    int heap_n, heap_top; /* heap_len / heap_max */
    uint8_t depth[ZO_HEAP];
    uint16_t per_len[16]; /* bl_count */
} zo_huff;

/* the ordering of src/trees.c:377-379: lower frequency first, then shallower subtree;
 * when both tie, the FIRST argument is "smaller" -- heap mechanics decide. */
static inline int zo_before(const zo_tree *t, const uint8_t *depth, int a, int b)
{
    return t->freq[a] < t->freq[b] || (t->freq[a] == t->freq[b] && depth[a] <= depth[b]);
}

/* pqdownheap, src/trees.c:387-414 */
static void zo_sift(zo_huff *h, const zo_tree *t, int k)
{
    int v = h->heap[k];
    for (int j = k << 1; j <= h->heap_n; j <<= 1) {
        if (j < h->heap_n && zo_before(t, h->depth, h->heap[j + 1], h->heap[j]))
            j++;
        if (zo_before(t, h->depth, v, h->heap[j]))
            break;
        h->heap[k] = h->heap[j];
        k = j;
    }
    h->heap[k] = v;
}

/* gen_bitlen, src/trees.c:426-508 */
static void zo_assign_lengths(zo_huff *h, zo_tree *t, const zo_tree_kind *kind)
{
    int over = 0;
    memset(h->per_len, 0, sizeof h->per_len);
    t->len[h->heap[h->heap_top]] = 0;
    int i;
    for (i = h->heap_top + 1; i < ZO_HEAP; i++) {
        int n = h->heap[i];
        int bits = t->len[t->parent[n]] + 1;
        if (bits > kind->max_len) {
            bits = kind->max_len;
            over++; /* counted for internal nodes too (:457 precedes :461) */
        }
        t->len[n] = (uint16_t)bits;
        if (n > t->max_code)
            continue;
        h->per_len[bits]++;
        int xb = n >= kind->extra_from ? kind->extra[n - kind->extra_from] : 0;
        h->opt_bits += (uint32_t)t->freq[n] * (uint32_t)(bits + xb);
        if (kind->has_static)
            h->static_bits +=
                (uint32_t)t->freq[n] * (uint32_t)((kind->is_dist ? 5 : (int)zo_static_llen((uint32_t)n)) + xb);
    }
    if (over == 0)
        return;
    do { /* :477-489 */
        int bits = kind->max_len - 1;
        while (h->per_len[bits] == 0)
            bits--;
        h->per_len[bits]--;
        h->per_len[bits + 1] += 2;
        h->per_len[kind->max_len]--;
        over -= 2;
    } while (over > 0);
    for (int bits = kind->max_len; bits != 0; bits--) { /* :496-507, i == ZO_HEAP here */
        int n = h->per_len[bits];
        while (n != 0) {
            int m = h->heap[--i];
            if (m > t->max_code)
                continue;
            if (t->len[m] != (uint16_t)bits) {
                h->opt_bits += ((uint32_t)bits - t->len[m]) * t->freq[m];
                t->len[m] = (uint16_t)bits;
            }
            n--;
        }
    }
}

/* gen_codes, src/trees.c:518-549 */
static void zo_assign_codes(zo_tree *t, const uint16_t *per_len)
{
    uint16_t next[16];
    uint32_t code = 0;
    for (int b = 1; b <= 15; b++) {
        code = (code + per_len[b - 1]) << 1;
        next[b] = (uint16_t)code;
    }
    for (int n = 0; n <= t->max_code; n++) {
        int l = t->len[n];
        if (l)
            t->code[n] = (uint16_t)zo_bitrev(next[l]++, l);
    }
}

/* build_tree, src/trees.c:559-652 */
static void zo_build(zo_huff *h, zo_tree *t, const zo_tree_kind *kind)
{
    int top = -1;
    h->heap_n = 0;
    h->heap_top = ZO_HEAP;
    for (int n = 0; n < kind->elems; n++) {
        if (t->freq[n]) {
            h->heap[++h->heap_n] = top = n;
            h->depth[n] = 0;
        } else {
            t->len[n] = 0;
        }
    }
    while (h->heap_n < 2) { /* :595-610: force two codes */
        int node = top < 2 ? ++top : 0;
        h->heap[++h->heap_n] = node;
        t->freq[node] = 1;
        h->depth[node] = 0;
        h->opt_bits--;
        if (kind->has_static)
            h->static_bits -= kind->is_dist ? 5u : zo_static_llen((uint32_t)node);
    }
    t->max_code = top;
    for (int n = h->heap_n / 2; n >= 1; n--)
        zo_sift(h, t, n);

    int node = kind->elems;
    do { /* :624-640 */
        int a = h->heap[1];
        h->heap[1] = h->heap[h->heap_n--];
        zo_sift(h, t, 1);
        int b = h->heap[1];
        h->heap[--h->heap_top] = a;
        h->heap[--h->heap_top] = b;
        t->freq[node] = (uint16_t)(t->freq[a] + t->freq[b]);
        h->depth[node] = (uint8_t)((h->depth[a] >= h->depth[b] ? h->depth[a] : h->depth[b]) + 1);
        t->parent[a] = t->parent[b] = (uint16_t)node;
        h->heap[1] = node++;
        zo_sift(h, t, 1);
    } while (h->heap_n >= 2);
    h->heap[--h->heap_top] = h->heap[1];

    zo_assign_lengths(h, t, kind);
    zo_assign_codes(t, h->per_len);
}

/* scan_tree, src/trees.c:658-707: run-length statistics of one code-length array */
static void zo_scan_lengths(zo_huff *h, zo_tree *t)
{
    int prev = -1, next = t->len[0], count = 0, hi = 7, lo = 4;
    if (next == 0) {
        hi = 138;
        lo = 3;
    }
    t->len[t->max_code + 1] = 0xffff; /* guard, :674 */
    for (int n = 0; n <= t->max_code; n++) {
        int cur = next;
        next = t->len[n + 1];
        if (++count < hi && cur == next)
            continue;
        if (count < lo)
            h->bt.freq[cur] += (uint16_t)count;
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

/* ------------------------------------------------------------------------- */
/* stage E: bit packing                                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint8_t *out;
    uint32_t pos, cap;
    uint64_t acc;
    int fill;
    int overflow;
} zo_bits;

static inline void zo_byte(zo_bits *w, uint8_t b)
{
    if (w->pos < w->cap)
        w->out[w->pos] = b;
    else
        w->overflow = 1;
    w->pos++;
}

/* How many bytes the reference can hand out at this point: flush_pending() first moves
 * every complete byte of bi_buf to the pending buffer (_tr_flush_bits, src/deflate.c:933),
 * so it is simply every whole byte written so far. */
static uint32_t zo_produced(const zo_bits *w)
{
    return w->pos;
}

/* send_bits, src/trees.c:292-304: LSB-first */
static inline void zo_put(zo_bits *w, uint32_t value, int nbits)
{
    w->acc |= (uint64_t)value << w->fill;
    w->fill += nbits;
    while (w->fill >= 8) {
        zo_byte(w, (uint8_t)w->acc);
        w->acc >>= 8;
        w->fill -= 8;
    }
}

static inline void zo_align(zo_bits *w) /* bi_windup, src/trees.c:1081-1092 */
{
    if (w->fill > 0)
        zo_byte(w, (uint8_t)w->acc);
    w->acc = 0;
    w->fill = 0;
}

/* send_tree, src/trees.c:713-773 */
static void zo_send_lengths(zo_bits *w, const zo_huff *h, const zo_tree *t, int max_code)
{
    int prev = -1, next = t->len[0], count = 0, hi = 7, lo = 4;
    if (next == 0) {
        hi = 138;
        lo = 3;
    }
    for (int n = 0; n <= max_code; n++) {
        int cur = next;
        next = t->len[n + 1];
        if (++count < hi && cur == next)
            continue;
        if (count < lo) {
            do
                zo_put(w, h->bt.code[cur], h->bt.len[cur]);
            while (--count != 0);
        } else if (cur != 0) {
            if (cur != prev) {
                zo_put(w, h->bt.code[cur], h->bt.len[cur]);
                count--;
            }
            zo_put(w, h->bt.code[16], h->bt.len[16]);
            zo_put(w, (uint32_t)(count - 3), 2);
        } else if (count <= 10) {
            zo_put(w, h->bt.code[17], h->bt.len[17]);
            zo_put(w, (uint32_t)(count - 3), 3);
        } else {
            zo_put(w, h->bt.code[18], h->bt.len[18]);
            zo_put(w, (uint32_t)(count - 11), 7);
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

/* compress_block, src/trees.c:948-993.  fixed != 0 uses the static code. */
static void zo_send_symbols(zo_bits *w, const zo_huff *h, const zo_symbol *syms, uint32_t count,
                            int fixed)
{
    for (uint32_t i = 0; i < count; i++) {
        uint32_t d = syms[i].dist, lc = syms[i].lc;
        if (d == 0) {
            if (fixed)
                zo_put(w, zo_bitrev(lc < 144 ? 0x30 + lc : 0x190 + (lc - 144), lc < 144 ? 8 : 9),
                       lc < 144 ? 8 : 9);
            else
                zo_put(w, h->lt.code[lc], h->lt.len[lc]);
            continue;
        }
        uint32_t c = zo_len_code(lc), sym = c + 257;
        if (fixed) {
            if (sym < 280)
                zo_put(w, zo_bitrev(sym - 256, 7), 7);
            else
                zo_put(w, zo_bitrev(0xC0 + (sym - 280), 8), 8);
        } else
            zo_put(w, h->lt.code[sym], h->lt.len[sym]);
        if (zo_len_extra[c])
            zo_put(w, lc - zo_len_base[c], zo_len_extra[c]);
        d--;
        c = zo_dist_code(d);
        if (fixed)
            zo_put(w, zo_bitrev(c, 5), 5);
        else
            zo_put(w, h->dt.code[c], h->dt.len[c]);
        if (zo_dist_extra[c])
            zo_put(w, d - zo_dist_base[c], zo_dist_extra[c]);
    }
    if (fixed)
        zo_put(w, 0, 7); /* END_BLOCK, static code 0000000 */
    else
        zo_put(w, h->lt.code[256], h->lt.len[256]);
}

/* _tr_flush_block, src/trees.c:874-941 for level > 0 */
static void zo_emit_block(zo_bits *w, const uint8_t *in, const zo_symbol *syms, const zo_block *b,
                          int strategy)
{
    zo_huff h;
    memset(&h, 0, sizeof h);
    const zo_symbol *s = syms + b->sym_begin;
    h.lt.freq[256] = 1; /* init_block, :353 */
    for (uint32_t i = 0; i < b->sym_count; i++) {
        if (s[i].dist == 0)
            h.lt.freq[s[i].lc]++;
        else {
            h.lt.freq[257 + zo_len_code(s[i].lc)]++;
            h.dt.freq[zo_dist_code((uint32_t)s[i].dist - 1)]++;
        }
    }
    zo_build(&h, &h.lt, &zo_kind_l);
    zo_build(&h, &h.dt, &zo_kind_d);
    /* build_bl_tree, :779-806 */
    zo_scan_lengths(&h, &h.lt);
    zo_scan_lengths(&h, &h.dt);
    zo_build(&h, &h.bt, &zo_kind_bl);
    int last_bl;
    for (last_bl = ZO_BLCODES - 1; last_bl >= 3; last_bl--)
        if (h.bt.len[zo_bl_order[last_bl]] != 0)
            break;
    h.opt_bits += 3u * ((uint32_t)last_bl + 1) + 5 + 5 + 4;

    uint32_t opt_bytes = (h.opt_bits + 3 + 7) >> 3, static_bytes = (h.static_bits + 3 + 7) >> 3;
    if (static_bytes <= opt_bytes)
        opt_bytes = static_bytes;

    if (b->in_len + 4 <= opt_bytes && b->stored_ok) {
        /* _tr_stored_block, :838-849 */
        zo_put(w, (uint32_t)b->last, 3);
        zo_align(w);
        zo_byte(w, (uint8_t)b->in_len);
        zo_byte(w, (uint8_t)(b->in_len >> 8));
        zo_byte(w, (uint8_t)~b->in_len);
        zo_byte(w, (uint8_t)(~b->in_len >> 8));
        for (uint32_t i = 0; i < b->in_len; i++)
            zo_byte(w, in[b->in_begin + i]);
    } else if (strategy == 4 || static_bytes == opt_bytes) {
        zo_put(w, 2u + b->last, 3);
        zo_send_symbols(w, &h, s, b->sym_count, 1);
    } else {
        zo_put(w, 4u + b->last, 3);
        /* send_all_trees, :813-833 */
        zo_put(w, (uint32_t)(h.lt.max_code + 1 - 257), 5);
        zo_put(w, (uint32_t)(h.dt.max_code + 1 - 1), 5);
        zo_put(w, (uint32_t)(last_bl + 1 - 4), 4);
        for (int r = 0; r <= last_bl; r++)
            zo_put(w, h.bt.len[zo_bl_order[r]], 3);
        zo_send_lengths(w, &h, &h.lt, h.lt.max_code);
        zo_send_lengths(w, &h, &h.dt, h.dt.max_code);
        zo_send_symbols(w, &h, s, b->sym_count, 0);
    }
    if (b->last)
        zo_align(w);
}

/* ------------------------------------------------------------------------- */
/* whole-call compress                                                       */
/* ------------------------------------------------------------------------- */

#define ZO_DEFLATE_STATE_BYTES 5920u /* sizeof(deflate_state), LP64 build of the reference */
#define ZO_INFLATE_STATE_BYTES 7152u /* sizeof(inflate_state), LP64 build of the reference */

/* zsc_compress with source_len > max_block_len (src/zsc_compress.c:121-138): the wrapper
 * hands deflate() the input in sections of max_block_len with Z_FULL_FLUSH between them --
 * and the OUTPUT in slices of max_block_len too, refilling whichever of the two has run
 * out before each call.  Where a section ends the stream gets the empty stored block
 * 00 00 FF FF and the history is forgotten (src/deflate.c:1240-1252), UNLESS the output
 * slice ran out while the section's last block was being flushed: then deflate() returns
 * with the flush unfinished, the wrapper refills the input as well, and the next section is
 * compressed with the history, no marker (SURVEY finding 2).  A slice that runs out in the
 * middle of a section, when all of the section has already been read into the window, lets
 * the next section in early in the same way.  So what is in the stream depends on where the
 * compressed bytes fall relative to multiples of max_block_len; this function follows the
 * calls one by one. */
typedef struct {
    zo_parser z;
    zo_bits w;
    const uint8_t *source;
    uint32_t source_len;
    uint32_t run_abs;      /* where the current parser's position 0 lies in source */
    uint32_t given;        /* input handed to deflate() so far (absolute) */
    uint32_t delivered;    /* total_out */
    uint32_t avail_out;
    int wrap, level, strategy, wb;
    int header_done, finishing, trailer_done;
    uint32_t emitted_blocks; /* blocks of z.blocks already written to w */
    /* level 0 (deflate_stored, src/deflate.c:1679-1880): positions in the window and in the input */
    uint32_t st_strstart, st_block_start; /* s->strstart, s->block_start */
    uint32_t st_read;                     /* input bytes taken from next_in so far (absolute) */
    uint32_t st_emit;                     /* input bytes written into stored blocks so far (absolute) */
    uint32_t pending_buf_size;
} zo_stream;

static void zo_s_flush_pending(zo_stream *m)
{
    uint32_t have = zo_produced(&m->w) - m->delivered;
    uint32_t len = have < m->avail_out ? have : m->avail_out;
    m->delivered += len;
    m->avail_out -= len;
}

static void zo_s_emit_new_blocks(zo_stream *m)
{
    for (; m->emitted_blocks < m->z.nblocks; m->emitted_blocks++)
        zo_emit_block(&m->w, m->z.in, m->z.syms, &m->z.blocks[m->emitted_blocks], m->strategy);
}

/* one stored block: 3 header bits, byte alignment, LEN, NLEN, the bytes (src/trees.c:838-849) */
static void zo_s_stored_block(zo_stream *m, uint32_t len, int last)
{
    zo_put(&m->w, (uint32_t)last, 3);
    zo_align(&m->w);
    zo_byte(&m->w, (uint8_t)len);
    zo_byte(&m->w, (uint8_t)(len >> 8));
    zo_byte(&m->w, (uint8_t)~len);
    zo_byte(&m->w, (uint8_t)(~len >> 8));
    for (uint32_t i = 0; i < len; i++)
        zo_byte(&m->w, m->source[m->st_emit + i]);
    m->st_emit += len;
}

/* deflate_stored, src/deflate.c:1679-1880, for level 0.  Returns 0 need_more, 1 block_done,
 * 2 finish_started, 3 finish_done.  Block lengths depend on the output space of the moment,
 * so this follows the reference's arithmetic step by step; which bytes go where is simple:
 * the input, in order, st_emit being the next byte to be stored. */
static int zo_s_deflate_stored(zo_stream *m, int finish)
{
    const uint32_t w_size = m->z.wsize, window_size = 2u * w_size;
    uint32_t min_block = m->pending_buf_size - 5u < w_size ? m->pending_buf_size - 5u : w_size;
    uint32_t avail_in = m->given - m->st_read;
    const uint32_t used0 = avail_in;
    uint32_t len, left, have;
    int last = 0;
    do {
        len = 65535u;
        have = 5u; /* (bi_valid + 42) >> 3 with an empty bit buffer */
        if (m->avail_out < have)
            break;
        have = m->avail_out - have;
        left = m->st_strstart - m->st_block_start;
        if (len > left + avail_in)
            len = left + avail_in;
        if (len > have)
            len = have;
        if (len < min_block && ((len == 0 && !finish) || len != left + avail_in))
            break; /* flush is never Z_NO_FLUSH here */
        last = finish && len == left + avail_in;
        /* header through the pending buffer, bytes straight to next_out (:1745-1775) */
        uint32_t from_window = left < len ? left : len;
        zo_s_stored_block(m, len, last);
        m->st_block_start += from_window;
        m->st_read += len - from_window;
        avail_in -= len - from_window;
        m->delivered += 5u + len;
        m->avail_out -= 5u + len;
    } while (!last);

    uint32_t used = used0 - avail_in; /* input bytes copied directly */
    if (used) {
        if (used >= w_size) {
            m->st_strstart = w_size;
        } else {
            if (window_size - m->st_strstart <= used)
                m->st_strstart -= w_size;
            m->st_strstart += used;
        }
        m->st_block_start = m->st_strstart;
    }
    if (last)
        return 3;
    if (!finish && avail_in == 0 && m->st_strstart == m->st_block_start)
        return 1;

    /* fill the window with any remaining input (:1822-1843) */
    have = window_size - m->st_strstart - 1u;
    if (avail_in > have && m->st_block_start >= w_size) {
        m->st_block_start -= w_size;
        m->st_strstart -= w_size;
        have += w_size;
    }
    if (have > avail_in)
        have = avail_in;
    if (have) {
        m->st_read += have;
        avail_in -= have;
        m->st_strstart += have;
    }
    /* a stored block through the pending buffer, if worth it or flushing (:1850-1876) */
    have = m->pending_buf_size - 5u < 65535u ? m->pending_buf_size - 5u : 65535u;
    min_block = have < w_size ? have : w_size;
    left = m->st_strstart - m->st_block_start;
    if (left >= min_block || ((left || finish) && avail_in == 0 && left <= have)) {
        len = left < have ? left : have;
        last = finish && avail_in == 0 && len == left;
        zo_s_stored_block(m, len, last);
        m->st_block_start += len;
        zo_s_flush_pending(m);
    }
    return last ? 2 : 0;
}

/* one deflate() call; returns ZO_OK, 1 (Z_STREAM_END) or ZO_BUF_ERROR */
static int zo_s_deflate(zo_stream *m, int finish)
{
    zo_parser *z = &m->z;
    if (m->avail_out == 0)
        return ZO_BUF_ERROR; /* src/deflate.c:987-990 */
    if (zo_produced(&m->w) != m->delivered) { /* :996-1008 */
        zo_s_flush_pending(m);
        if (m->avail_out == 0)
            return ZO_OK;
    }
    if (!m->header_done) { /* :1029-1090 */
        m->header_done = 1;
        if (m->wrap == 1) {
            uint32_t hdr = (8u + (((uint32_t)m->wb - 8u) << 4)) << 8;
            uint32_t lf = (m->strategy >= 2 || m->level < 2) ? 0u : m->level < 6 ? 1u : m->level == 6 ? 2u : 3u;
            hdr |= lf << 6;
            hdr += 31 - hdr % 31;
            zo_byte(&m->w, (uint8_t)(hdr >> 8));
            zo_byte(&m->w, (uint8_t)hdr);
        } else if (m->wrap == 2) {
            static const uint8_t fixed[8] = {31, 139, 8, 0, 0, 0, 0, 0};
            for (int i = 0; i < 8; i++)
                zo_byte(&m->w, fixed[i]);
            zo_byte(&m->w, m->level == 9 ? 2 : (m->strategy >= 2 || m->level < 2) ? 4 : 0);
            zo_byte(&m->w, 3);
        }
        if (m->wrap) {
            zo_s_flush_pending(m);
            if (zo_produced(&m->w) != m->delivered)
                return ZO_OK;
        }
    }
    if (!m->finishing && m->level == 0) { /* :1211-1260 with deflate_stored */
        int bs = zo_s_deflate_stored(m, finish);
        if (bs == 2 || bs == 3)
            m->finishing = 1;
        if (bs == 0 || bs == 2)
            return ZO_OK;
        if (bs == 1) { /* block_done: the flush marker; strstart = 0 as lookahead is 0 */
            zo_put(&m->w, 0, 3);
            zo_align(&m->w);
            zo_byte(&m->w, 0);
            zo_byte(&m->w, 0);
            zo_byte(&m->w, 0xff);
            zo_byte(&m->w, 0xff);
            m->st_strstart = m->st_block_start = 0;
            zo_s_flush_pending(m);
            return ZO_OK;
        }
    } else if (!m->finishing) { /* :1211-1260 */
        z->n = m->given - m->run_abs;
        for (;;) {
            int ev = zo_run(z);
            if (ev == ZO_EV_BLOCK) {
                zo_s_emit_new_blocks(m); /* FLUSH_BLOCK(s, 0) */
                zo_s_flush_pending(m);
                if (m->avail_out == 0)
                    return ZO_OK; /* need_more */
                continue;
            }
            /* the input given so far is used up */
            zo_end_of_input(z);
            if (finish) {
                zo_cut_block(z, z->p, 1);
                zo_s_emit_new_blocks(m);
                zo_s_flush_pending(m);
                m->finishing = 1; /* FINISH_STATE */
                if (m->avail_out == 0)
                    return ZO_OK; /* finish_started */
                break;
            }
            if (z->nsyms != z->block_begin_sym) { /* if (s->last_lit) FLUSH_BLOCK(s, 0) */
                zo_cut_block(z, z->p, 0);
                zo_s_emit_new_blocks(m);
                zo_s_flush_pending(m);
                if (m->avail_out == 0)
                    return ZO_OK; /* need_more: the flush marker is never written (finding 2) */
            }
            /* block_done with Z_FULL_FLUSH: _tr_stored_block(s, 0, 0, 0), forget the history */
            zo_put(&m->w, 0, 3);
            zo_align(&m->w);
            zo_byte(&m->w, 0);
            zo_byte(&m->w, 0);
            zo_byte(&m->w, 0xff);
            zo_byte(&m->w, 0xff);
            m->run_abs += z->p;
            zo_parser_restart(z, m->source + m->run_abs);
            zo_s_flush_pending(m);
            return ZO_OK; /* whether or not avail_out is 0 (:1253-1257, :1262-1264) */
        }
    }
    if (!finish)
        return ZO_OK;
    if (m->wrap == 0)
        return 1;
    if (!m->trailer_done) { /* :1270-1290 */
        m->trailer_done = 1;
        if (m->wrap == 1) {
            uint32_t a = zo_adler32(1u, m->source, m->source_len);
            zo_byte(&m->w, (uint8_t)(a >> 24));
            zo_byte(&m->w, (uint8_t)(a >> 16));
            zo_byte(&m->w, (uint8_t)(a >> 8));
            zo_byte(&m->w, (uint8_t)a);
        } else {
            uint32_t c = zo_crc32(0u, m->source, m->source_len);
            for (int i = 0; i < 4; i++)
                zo_byte(&m->w, (uint8_t)(c >> (8 * i)));
            for (int i = 0; i < 4; i++)
                zo_byte(&m->w, (uint8_t)(m->source_len >> (8 * i)));
        }
        zo_s_flush_pending(m);
        return zo_produced(&m->w) != m->delivered ? ZO_OK : 1;
    }
    return 1; /* wrap was negated after the trailer (:1291-1294) */
}

static int zo_compress_sections(uint8_t *dest, uint32_t *dest_len, const uint8_t *source,
                                uint32_t source_len, uint32_t max_block_len, int level, int wrap,
                                int wb, int mem_level, int strategy, uint32_t bound)
{
    const uint32_t cap_in = *dest_len;
    zo_stream *m = (zo_stream *)calloc(1, sizeof *m);
    /* the stream can outgrow the wrapper's own bound (5 bytes per flush marker against the
     * 4 it allows for); the caller then gets Z_BUF_ERROR from a dest of that size, like here */
    uint32_t scratch_cap = bound + source_len / max_block_len * 8u + (source_len >> 3) + 4096u;
    uint8_t *scratch = (uint8_t *)malloc(scratch_cap);
    zo_symbol *syms = (zo_symbol *)malloc(((size_t)source_len + 1) * sizeof(zo_symbol));
    const uint32_t sym_cap = (1u << (mem_level + 6)) - 1u;
    uint32_t max_blocks = source_len / sym_cap + source_len / max_block_len + 4;
    zo_block *blocks = (zo_block *)malloc((size_t)max_blocks * sizeof(zo_block));
    zo_parser *z = &m->z;
    z->wsize = 1u << wb;
    z->wmask = z->wsize - 1;
    z->max_dist = z->wsize - ZO_MIN_LOOKAHEAD;
    uint32_t hbits = (uint32_t)mem_level + 7u;
    z->hmask = (1u << hbits) - 1;
    z->hshift = (hbits + 2u) / 3u;
    z->sym_cap = sym_cap;
    z->cfg = zo_levels[level];
    z->strategy = strategy;
    z->head = (uint32_t *)calloc((size_t)z->hmask + 1, sizeof(uint32_t));
    z->link = (uint32_t *)calloc(z->wsize, sizeof(uint32_t));
    z->syms = syms;
    z->blocks = blocks;
    if (!m || !scratch || !syms || !blocks || !z->head || !z->link) {
        free(scratch);
        free(syms);
        free(blocks);
        free(z->head);
        free(z->link);
        free(m);
        return ZO_MEM_ERROR;
    }
    zo_parser_restart(z, source);
    m->w.out = scratch;
    m->w.cap = scratch_cap;
    m->source = source;
    m->source_len = source_len;
    m->wrap = wrap;
    m->level = level;
    m->strategy = strategy;
    m->wb = wb;
    m->pending_buf_size = (1u << (mem_level + 6)) * 4u; /* lit_bufsize * (sizeof(U16) + 2), :362 */

    /* the wrapper's loop, src/zsc_compress.c:121-138 */
    uint32_t left_dest = cap_in, left_src = source_len;
    int err = ZO_OK;
    while (err == ZO_OK) {
        if (m->avail_out == 0) {
            m->avail_out = left_dest < max_block_len ? left_dest : max_block_len;
            left_dest -= m->avail_out;
        }
        /* avail_in == 0: everything given so far has been read into the window */
        if (level == 0 ? m->st_read == m->given : m->run_abs + z->data_end == m->given) {
            uint32_t take = left_src < max_block_len ? left_src : max_block_len;
            m->given += take;
            left_src -= take;
        }
        err = zo_s_deflate(m, left_src == 0);
    }
    uint32_t give = m->delivered;
    if (!m->w.overflow)
        memcpy(dest, scratch, give);
    *dest_len = give;
    int rc = err == 1 ? ZO_OK : err;
    if (m->w.overflow)
        rc = ZO_MEM_ERROR;
    free(scratch);
    free(syms);
    free(blocks);
    free(z->head);
    free(z->link);
    free(m);
    return rc;
}


int zo_compress(uint8_t *dest, uint32_t *dest_len, const uint8_t *source, uint32_t source_len,
                uint32_t max_block_len, uint32_t work_len, int level, int window_bits,
                int mem_level, int strategy, int *unsupported)
{
    uint32_t cap_in = *dest_len;
    *dest_len = 0;
    *unsupported = 0;

    /* src/zsc_compress.c:74-88 */
    uint32_t need;
    int err = zo_compress_work_size(window_bits, mem_level, ZO_DEFLATE_STATE_BYTES, &need);
    if (err != ZO_OK)
        return err;
    if (work_len < need)
        return ZO_MEM_ERROR;

    /* deflateInit2_, src/deflate.c:305-331 */
    if (level == -1)
        level = 6;
    int wb, wrap = zo_split_wbits(window_bits, &wb);
    if (mem_level < 1 || mem_level > 9 || wb < 8 || wb > 15 || level < 0 || level > 9 ||
        strategy < 0 || strategy > 4 || (wb == 8 && wrap != 1))
        return ZO_STREAM_ERROR;
    if (wb == 8)
        wb = 9;

    if (max_block_len == 0) {
        *unsupported = 1;
        return ZO_STREAM_ERROR;
    }

    uint32_t bound;
    err = zo_compress_max_output(source_len, max_block_len, level, window_bits, mem_level, &bound);
    if (err != ZO_OK)
        return err;
    if (source_len > max_block_len || level == 0) { /* level 0: block sizes follow the output slices */
        *dest_len = cap_in;
        return zo_compress_sections(dest, dest_len, source, source_len, max_block_len, level, wrap, wb,
                                    mem_level, strategy, bound);
    }

    /* stages P, H+E into a scratch stream of the worst-case size */
    uint32_t scratch_cap = bound + 64;
    uint8_t *scratch = (uint8_t *)malloc(scratch_cap);
    zo_symbol *syms = (zo_symbol *)malloc(((size_t)source_len + 1) * sizeof(zo_symbol));
    uint32_t max_blocks = source_len / ((1u << (mem_level + 6)) - 1u) + 2;
    zo_block *blocks = (zo_block *)malloc((size_t)max_blocks * sizeof(zo_block));
    if (!scratch || !syms || !blocks) {
        free(scratch);
        free(syms);
        free(blocks);
        return ZO_MEM_ERROR;
    }
    uint32_t nsyms = 0, nblocks = 0;
    err = zo_parse(source, source_len, level, wb, mem_level, strategy, syms, &nsyms, blocks, &nblocks);
    if (err != ZO_OK) {
        free(scratch);
        free(syms);
        free(blocks);
        return err;
    }

    zo_bits w = {scratch, 0, scratch_cap, 0, 0, 0};
    if (wrap == 1) {
        /* src/deflate.c:1029-1049 */
        uint32_t hdr = (8u + (((uint32_t)wb - 8u) << 4)) << 8;
        uint32_t lf = (strategy >= 2 || level < 2) ? 0u : level < 6 ? 1u : level == 6 ? 2u : 3u;
        hdr |= lf << 6;
        hdr += 31 - hdr % 31;
        zo_byte(&w, (uint8_t)(hdr >> 8));
        zo_byte(&w, (uint8_t)hdr);
    } else if (wrap == 2) {
        /* src/deflate.c:1066-1082 */
        static const uint8_t fixed[8] = {31, 139, 8, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8; i++)
            zo_byte(&w, fixed[i]);
        zo_byte(&w, level == 9 ? 2 : (strategy >= 2 || level < 2) ? 4 : 0);
        zo_byte(&w, 3); /* OS_CODE, include/zsc/zutil.h:133-135 */
    }
    for (uint32_t i = 0; i < nblocks; i++)
        zo_emit_block(&w, source, syms, &blocks[i], strategy);
    if (wrap == 1) {
        uint32_t a = zo_adler32(1u, source, source_len);
        zo_byte(&w, (uint8_t)(a >> 24));
        zo_byte(&w, (uint8_t)(a >> 16));
        zo_byte(&w, (uint8_t)(a >> 8));
        zo_byte(&w, (uint8_t)a);
    } else if (wrap == 2) {
        uint32_t c = zo_crc32(0u, source, source_len);
        for (int i = 0; i < 4; i++)
            zo_byte(&w, (uint8_t)(c >> (8 * i)));
        for (int i = 0; i < 4; i++)
            zo_byte(&w, (uint8_t)(source_len >> (8 * i)));
    }
    free(syms);
    free(blocks);
    if (w.overflow) { /* cannot happen: the bound is an upper bound */
        free(scratch);
        return ZO_MEM_ERROR;
    }

    /* the reference hands out dest in slices and stops with Z_BUF_ERROR once it is
     * used up (src/zsc_compress.c:126-140, src/deflate.c:987-990): the caller gets
     * the prefix that fitted. */
    uint32_t give = w.pos <= cap_in ? w.pos : cap_in;
    memcpy(dest, scratch, give);
    *dest_len = give;
    free(scratch);
    return w.pos <= cap_in ? ZO_OK : ZO_BUF_ERROR;
}

/* ------------------------------------------------------------------------- */
/* whole-call uncompress                                                     */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *src;
    uint32_t n, pos;
    uint64_t acc;
    uint32_t fill;
} zo_reader;

/* make sure `need` bits are buffered; 0 when the input ends first */
static inline int zo_want(zo_reader *r, uint32_t need)
{
    while (r->fill < need) {
        if (r->pos >= r->n)
            return 0;
        r->acc |= (uint64_t)r->src[r->pos++] << r->fill;
        r->fill += 8;
    }
    return 1;
}

static inline uint32_t zo_take(zo_reader *r, uint32_t nbits)
{
    uint32_t v = (uint32_t)(r->acc & ((1ull << nbits) - 1));
    r->acc >>= nbits;
    r->fill -= nbits;
    return v;
}

/* canonical code described by its lengths: first code / first symbol index per length */
typedef struct {
    uint16_t count[16];
    uint16_t symbol[288];
    int max_len;
    int empty;
} zo_code;

/* validity rules of inflate_table, src/inftrees.c:130-177: over-subscribed sets are
 * rejected; incomplete sets are rejected except a lone 1-bit code in a literal/length
 * or distance set (:175). kind: 0 code lengths, 1 literal/length, 2 distance. */
static int zo_code_build(zo_code *c, const uint16_t *lens, int n, int kind)
{
    memset(c->count, 0, sizeof c->count);
    for (int i = 0; i < n; i++)
        c->count[lens[i]]++;
    int max = 15;
    while (max >= 1 && c->count[max] == 0)
        max--;
    c->max_len = max;
    c->empty = max == 0;
    if (max == 0)
        return 0; /* decoding reports the error when a symbol is needed (:150-158) */
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= c->count[l];
        if (left < 0)
            return -1;
    }
    if (left > 0 && (kind == 0 || max != 1))
        return -1;
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++)
        offs[l + 1] = (uint16_t)(offs[l] + c->count[l]);
    for (int i = 0; i < n; i++)
        if (lens[i])
            c->symbol[offs[lens[i]]++] = (uint16_t)i;
    return 0;
}

/* decode one symbol: -1 input exhausted, -2 code not in the set */
static int zo_decode(zo_reader *r, const zo_code *c)
{
    if (c->empty) {
        /* a table of invalid-code markers with bits = 1 (src/inftrees.c:150-158) */
        if (!zo_want(r, 1))
            return -1;
        (void)zo_take(r, 1); /* DROPBITS(here.bits) comes before the op test, src/inflate.c:1217-1236 */
        return -2;
    }
    uint32_t code = 0, first = 0, index = 0;
    for (int l = 1; l <= c->max_len; l++) {
        if (!zo_want(r, (uint32_t)l))
            return -1;
        code |= (uint32_t)((r->acc >> (l - 1)) & 1);
        uint32_t cnt = c->count[l];
        if (code < first + cnt) {
            (void)zo_take(r, (uint32_t)l);
            return c->symbol[index + (code - first)];
        }
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    /* only reachable for the lone-1-bit-code case: the other 1-bit pattern */
    (void)zo_take(r, (uint32_t)c->max_len);
    return -2;
}

static const uint16_t zo_inf_lbase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                          31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t zo_inf_dbase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                          33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                          1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};

/* syncsearch, src/inflate.c:1523-1545: advance the 00 00 FF FF matcher over buf */
static uint32_t zo_syncsearch(uint32_t *have, const uint8_t *buf, uint32_t len)
{
    uint32_t got = *have, next = 0;
    while (next < len && got < 4) {
        if (buf[next] == (got < 2 ? 0 : 0xff))
            got++;
        else if (buf[next])
            got = 0;
        else
            got = 4 - got;
        next++;
    }
    *have = got;
    return next;
}

int zo_uncompress(uint8_t *dest, uint32_t *dest_len, const uint8_t *source, uint32_t *source_len,
                  uint32_t work_len, int window_bits)
{
    uint32_t cap = *dest_len, avail = *source_len;
    *dest_len = 0;
    *source_len = 0;

    uint32_t need;
    int err = zo_uncompress_work_size(window_bits, ZO_INFLATE_STATE_BYTES, &need);
    if (err != ZO_OK)
        return err;
    if (work_len < need)
        return ZO_MEM_ERROR;

    /* inflateReset2, src/inflate.c:341-356 */
    int wrap, wb = window_bits;
    if (wb < 0) {
        wrap = 0;
        wb = -wb;
    } else {
        wrap = (wb >> 4) + 5;
        if (wb < 48)
            wb &= 15;
    }
    if (wb && (wb < 8 || wb > 15))
        return ZO_STREAM_ERROR;

    zo_reader r = {source, avail, 0, 0, 0};
    uint32_t out = 0, dmax = 32768u;
    int data_errors = 0;
    uint32_t out_base = 0; /* output of the current inflate() call starts here: nothing before it
                              can be copied from once inflateSync has reset the window */
    int gzip = 0, rc = ZO_OK;

#define ZO_NEED(nb)          \
    if (!zo_want(&r, (nb))) { \
        rc = ZO_BUF_ERROR;    \
        goto done;            \
    }
/* A data error.  At every such point the bit buffer holds exactly what the reference's
 * `hold`/`bits` hold there (bytes are pulled one at a time as NEEDBITS does, and a field is
 * only taken out of the buffer where the reference drops it), because inflateSync starts
 * its search in those buffered bits (src/inflate.c:1570-1582). */
#define ZO_BAD goto bad;

    /* HEAD .. HCRC, src/inflate.c:740-954 */
    if (wrap) {
        ZO_NEED(16);
        uint32_t hw = (uint32_t)(r.acc & 0xffff);
        if ((wrap & 2) && hw == 0x8b1f) {
            gzip = 1;
            (void)zo_take(&r, 16);
            ZO_NEED(16);
            uint32_t flags = (uint32_t)(r.acc & 0xffff);
            if ((flags & 0xff) != 8)
                ZO_BAD;
            if (flags & 0xe000)
                ZO_BAD;
            (void)zo_take(&r, 16); /* INITBITS, :779 */
            ZO_NEED(32);
            (void)zo_take(&r, 32); /* mtime */
            ZO_NEED(16);
            (void)zo_take(&r, 16); /* xfl, os */
            int want_crc = (flags & 0x0200) && (wrap & 4);
            if (flags & 0x0400) {
                ZO_NEED(16);
                uint32_t xlen = zo_take(&r, 16);
                if (r.n - r.pos < xlen) {
                    r.pos = r.n;
                    rc = ZO_BUF_ERROR;
                    goto done;
                }
                r.pos += xlen;
            }
            if (flags & 0x0800) {
                for (;;) {
                    if (r.pos >= r.n) {
                        rc = ZO_BUF_ERROR;
                        goto done;
                    }
                    if (source[r.pos++] == 0)
                        break;
                }
            }
            if (flags & 0x1000) {
                for (;;) {
                    if (r.pos >= r.n) {
                        rc = ZO_BUF_ERROR;
                        goto done;
                    }
                    if (source[r.pos++] == 0)
                        break;
                }
            }
            if (flags & 0x0200) {
                uint32_t upto = r.pos;
                ZO_NEED(16);
                uint32_t got = (uint32_t)(r.acc & 0xffff);
                if (want_crc && got != (zo_crc32(0, source, upto) & 0xffff))
                    ZO_BAD;
                (void)zo_take(&r, 16);
            }
        } else {
            if (!(wrap & 1) || ((((hw & 0xff) << 8) + (hw >> 8)) % 31))
                ZO_BAD;
            if ((hw & 0xf) != 8)
                ZO_BAD;
            (void)zo_take(&r, 4); /* DROPBITS(4), :757 */
            uint32_t len = ((hw >> 4) & 0xf) + 8;
            uint32_t wbits_eff = wb ? (uint32_t)wb : len;
            if (len > 15 || len > wbits_eff)
                ZO_BAD;
            dmax = 1u << len;
            if (hw & 0x2000) {
                /* FDICT (bit 5 of FLG = bit 13 here): Z_NEED_DICT = 2 in the reference; the
                 * one-shot wrapper then fails with that code (src/zsc_uncompr.c:132-140) */
                (void)zo_take(&r, 12);
                ZO_NEED(32);
                (void)zo_take(&r, 32);
                /* the DICT state returns straight out of inflate() without the exit
                 * bookkeeping (src/inflate.c:961-965), so nothing counts as consumed */
                *dest_len = 0;
                *source_len = 0;
                return 2;
            }
            (void)zo_take(&r, 12);
        }
    }

    /* TYPE .. MATCH, src/inflate.c:975-1321 and src/inffast.c:125-297 */
blocks:
    for (;;) {
        ZO_NEED(3);
        uint32_t last = zo_take(&r, 1);
        uint32_t type = zo_take(&r, 2);
        if (type == 3)
            ZO_BAD;
        if (type == 0) {
            (void)zo_take(&r, r.fill & 7);
            ZO_NEED(32);
            uint32_t v = (uint32_t)(r.acc & 0xffffffffu);
            if ((v & 0xffff) != ((v >> 16) ^ 0xffff))
                ZO_BAD;
            (void)zo_take(&r, 32);
            uint32_t len = v & 0xffff;
            /* the bit buffer is empty here (INITBITS, :1019) */
            while (len) {
                if (r.pos >= r.n || out >= cap) {
                    rc = ZO_BUF_ERROR;
                    goto done;
                }
                dest[out++] = source[r.pos++];
                len--;
            }
        } else {
            zo_code lcode, dcode;
            if (type == 1) {
                uint16_t l[288];
                for (int i = 0; i < 288; i++)
                    l[i] = (uint16_t)zo_static_llen((uint32_t)i);
                (void)zo_code_build(&lcode, l, 288, 1);
                /* the fixed distance table has 32 five-bit entries, two invalid
                 * (src/inflate.c:122-206); a complete 5-bit code over 32 symbols */
                memset(dcode.count, 0, sizeof dcode.count);
                dcode.count[5] = 32;
                dcode.max_len = 5;
                dcode.empty = 0;
                for (int i = 0; i < 32; i++)
                    dcode.symbol[i] = (uint16_t)i;
            } else {
                ZO_NEED(14);
                uint32_t nlen = zo_take(&r, 5) + 257, ndist = zo_take(&r, 5) + 1,
                         ncode = zo_take(&r, 4) + 4;
                if (nlen > 286 || ndist > 30)
                    ZO_BAD;
                uint16_t lens[320];
                memset(lens, 0, sizeof lens);
                for (uint32_t i = 0; i < ncode; i++) {
                    ZO_NEED(3);
                    lens[zo_bl_order[i]] = (uint16_t)zo_take(&r, 3);
                }
                zo_code cl;
                if (zo_code_build(&cl, lens, 19, 0))
                    ZO_BAD;
                uint32_t have = 0;
                memset(lens, 0, sizeof lens);
                while (have < nlen + ndist) {
                    int sym = zo_decode(&r, &cl);
                    if (sym == -1) {
                        rc = ZO_BUF_ERROR;
                        goto done;
                    }
                    if (sym == -2)
                        sym = 0; /* an all-zero code-length code: the CODELENS state never looks at
                                    the invalid-code marker's op, only at its val 0 and bits 1
                                    (src/inflate.c:1105-1114, src/inftrees.c:150-158) */
                    if (sym < 16) {
                        lens[have++] = (uint16_t)sym;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        ZO_NEED(2);
                        if (have == 0)
                            ZO_BAD;
                        val = lens[have - 1];
                        rep = 3 + zo_take(&r, 2);
                    } else if (sym == 17) {
                        ZO_NEED(3);
                        rep = 3 + zo_take(&r, 3);
                    } else {
                        ZO_NEED(7);
                        rep = 11 + zo_take(&r, 7);
                    }
                    if (have + rep > nlen + ndist)
                        ZO_BAD;
                    while (rep--)
                        lens[have++] = (uint16_t)val;
                }
                if (lens[256] == 0)
                    ZO_BAD;
                if (zo_code_build(&lcode, lens, (int)nlen, 1))
                    ZO_BAD;
                if (zo_code_build(&dcode, lens + nlen, (int)ndist, 2))
                    ZO_BAD;
            }
            for (;;) {
                int sym = zo_decode(&r, &lcode);
                if (sym == -1) {
                    rc = ZO_BUF_ERROR;
                    goto done;
                }
                if (sym == -2)
                    ZO_BAD;
                if (sym < 256) {
                    if (out >= cap) {
                        rc = ZO_BUF_ERROR;
                        goto done;
                    }
                    dest[out++] = (uint8_t)sym;
                    continue;
                }
                if (sym == 256)
                    break;
                if (sym > 285)
                    ZO_BAD; /* 286, 287 of the fixed code */
                uint32_t c = (uint32_t)sym - 257, eb = zo_len_extra[c];
                ZO_NEED(eb);
                uint32_t len = zo_inf_lbase[c] + zo_take(&r, eb);
                int ds = zo_decode(&r, &dcode);
                if (ds == -1) {
                    rc = ZO_BUF_ERROR;
                    goto done;
                }
                if (ds == -2 || ds > 29)
                    ZO_BAD;
                eb = zo_dist_extra[ds];
                ZO_NEED(eb);
                uint32_t dist = zo_inf_dbase[ds] + zo_take(&r, eb);
                if (dist > dmax) /* DISTEXT, :1266-1272 */
                    ZO_BAD;
                if (out >= cap) { /* MATCH leaves on a full output before it looks at the distance (:1277) */
                    rc = ZO_BUF_ERROR;
                    goto done;
                }
                if (dist > out - out_base) /* :1279-1288 */
                    ZO_BAD;
                while (len--) {
                    if (out >= cap) {
                        rc = ZO_BUF_ERROR;
                        goto done;
                    }
                    dest[out] = dest[out - dist];
                    out++;
                }
            }
        }
        if (last)
            break;
    }

    /* CHECK / LENGTH, src/inflate.c:1322-1354 */
    (void)zo_take(&r, r.fill & 7);
    if (wrap) {
        ZO_NEED(32);
        uint32_t v = (uint32_t)(r.acc & 0xffffffffu);
        if (wrap & 4) {
            /* state->check runs on across an inflateSync: it covers everything written */
            uint32_t want = gzip ? zo_crc32(0, dest, out) : zo_adler32(1, dest, out);
            uint32_t got = gzip ? v : ((v >> 24) | ((v >> 8) & 0xff00) | ((v & 0xff00) << 8) | (v << 24));
            if (got != want)
                ZO_BAD;
        }
        (void)zo_take(&r, 32);
        if (gzip) {
            ZO_NEED(32);
            /* state->total restarts at an inflateSync (inflateResetKeep, :288) */
            if ((uint32_t)(r.acc & 0xffffffffu) != out - out_base)
                ZO_BAD;
            (void)zo_take(&r, 32);
        }
    }
    rc = 1; /* Z_STREAM_END */
    goto done;

bad:
    /* zsc_uncompress answers Z_DATA_ERROR with inflateSync (src/zsc_uncompr.c:109-125,
     * src/inflate.c:1547-1604): look for the next 00 00 FF FF -- the empty stored block a
     * full flush leaves between sections -- first in the buffered bits, then in the input,
     * and decode on from there as a raw stream with an empty window. */
    data_errors++;
    if (r.pos >= r.n && r.fill < 8) { /* :1562-1565 */
        rc = ZO_BUF_ERROR;
        goto done;
    }
    {
        uint32_t hold = (uint32_t)r.acc, bits = r.fill, have = 0, len = 0;
        uint8_t buf[4];
        hold <<= bits & 7; /* sic, :1571 */
        bits -= bits & 7;
        while (bits >= 8) {
            buf[len++] = (uint8_t)hold;
            hold >>= 8;
            bits -= 8;
        }
        (void)zo_syncsearch(&have, buf, len);
        r.pos += zo_syncsearch(&have, source + r.pos, r.n - r.pos);
        r.acc = 0; /* inflateReset */
        r.fill = 0;
        if (have != 4) {
            rc = ZO_DATA_ERROR;
            goto done;
        }
        out_base = out; /* whave = 0: copies cannot reach behind this call's output */
        dmax = 32768u;
        goto blocks; /* mode = TYPE */
    }

done:
#undef ZO_NEED
#undef ZO_BAD
    *dest_len = out;
    /* bytes are pulled into the bit buffer only on demand, as in the reference's
     * NEEDBITS/PULLBYTE states, so everything pulled counts as consumed */
    *source_len = r.pos;
    if (rc == 1) /* a stream that ended well after a resynchronisation still reports the error (:149-152) */
        return data_errors ? ZO_DATA_ERROR : ZO_OK;
    /* src/zsc_uncompr.c:132-141: Z_OK would become Z_STREAM_ERROR; inflate(Z_FINISH)
     * reports an unfinished stream as Z_BUF_ERROR (src/inflate.c:1400-1402) */
    return rc;
}
