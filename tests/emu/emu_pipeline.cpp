/*
 * emu_pipeline.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Compiles the kernel sources of zsc_amd/csrc with -DZSC_WAVE_EMU (wave.h), i.e.
 * runs the very same wavefront code lane by lane on the host, so the CPU test
 * suite (-m "not gpu") can check kernel logic against the oracle without a GPU.
 * Nothing here is shipped or reachable from libzsc_hip.so.
 */
#define ZSC_WAVE_EMU 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct uint4 { uint32_t x, y, z, w; };

#include "../../zsc_amd/csrc/hash_sort.h"
#include "../../zsc_amd/csrc/lz_parse.h"
/* event counters for tools/seg_stats.py: [0] batches, [1] long compares; per segment a
 * record (segment | redo flag, batches) is appended to g_sg_log */
extern "C" { unsigned long long g_sg_cnt[16]; unsigned g_sg_log[1 << 20]; unsigned g_sg_nlog; }
static unsigned long long g_sg_mark;
static int g_sg_in_fallback; /* [11]: candidate steps of walks done the reference's way after a staircase */
static inline void sg_count(int what, unsigned n)
{
    if (what == 8)
        g_sg_in_fallback = 1;
    if (what == 5)
        g_sg_in_fallback = 0; /* (a new search) */
    if (what == 0 && g_sg_in_fallback)
        g_sg_cnt[11] += n;
    if (what < 2 || what >= 4)
        g_sg_cnt[what] += n;
    else if (what == 2) {
        g_sg_mark = g_sg_cnt[0];
        if (g_sg_nlog + 2 <= (1u << 20))
            g_sg_log[g_sg_nlog] = n;
    } else if (g_sg_nlog + 2 <= (1u << 20)) {
        g_sg_log[g_sg_nlog + 1] = (unsigned)(g_sg_cnt[0] - g_sg_mark);
        g_sg_nlog += 2;
    }
}
#define SG_COUNT(what, n) sg_count(what, n)
#include "../../zsc_amd/csrc/lz_parse_seg.h"
#include "../../zsc_amd/csrc/match_table.h"
#include "../../zsc_amd/csrc/lz_parse_simple.h"
#include "../../zsc_amd/csrc/huff_plan.h"
#include "../../zsc_amd/csrc/bit_emit.h"
#include "../../zsc_amd/csrc/checksum.h"
#include "../../zsc_amd/csrc/inflate.h"
#include "../../zsc_amd/csrc/sections.h"
#include <cstdio>
#include <cstdlib>

static const ZdLevel kLevels[10] = {
    {0, 0, 0, 0, 0},       {4, 4, 8, 4, 0},       {4, 5, 16, 8, 0},     {4, 6, 32, 32, 0},
    {4, 4, 16, 16, 1},     {8, 16, 32, 32, 1},    {8, 16, 128, 128, 1}, {8, 32, 128, 256, 1},
    {32, 128, 258, 1024, 1}, {32, 258, 258, 4096, 1}};

static int g_wbits = 15, g_mem_level = 8; /* emu_set_params: the window_bits / mem_level of the next calls */
extern "C" void emu_set_params(int wbits, int mem_level)
{
    g_wbits = wbits;
    g_mem_level = mem_level;
}
static ZdLevel level_cfg(int level)
{
    ZdLevel c = kLevels[level];
    c.wsize = 1u << g_wbits;
    c.max_dist = c.wsize - ZD_MIN_LOOKAHEAD;
    c.sym_cap = (1u << (g_mem_level + 6)) - 1u;
    c.hbits = (uint32_t)g_mem_level + 7u;
    return c;
}

struct EmuChains {
    std::vector<uint8_t> in;
    uint32_t n, ntiles;
    std::vector<uint32_t> sorted, tmp;
    std::vector<uint16_t> rank, dir, hib;
    std::vector<uint32_t> cnt;
    std::vector<uint32_t> r2; /* the match table, when built */
};

static void build_chains(EmuChains &c, const uint8_t *src, uint32_t n)
{
    c.n = n;
    c.in.assign((size_t)n + 64, 0);
    memcpy(c.in.data(), src, n);
    c.ntiles = n == 0 ? 1 : (n + ZD_TILE - 1) / ZD_TILE;
    c.sorted.assign((size_t)c.ntiles * ZD_TILE, 0xdeadbeef);
    c.tmp.assign((size_t)c.ntiles * ZD_TILE, 0xdeadbeef);
    c.rank.assign((size_t)n + 64, 0xdead);
    c.dir.assign((size_t)c.ntiles * ZD_DIR_STRIDE, 0xdead);
    c.hib.assign((size_t)n + 64, 0xdead);
    c.cnt.assign((size_t)n + 64, 0xdeaddead);
    uint32_t owners = n >= 3 ? n - 2 : 0; /* positions 0..n-3 own a 3-byte string */
    for (uint32_t t = 0; t < c.ntiles; t++) {
        HsTile tile;
        tile.in = c.in.data();
        tile.n = n;
        tile.start = t * ZD_TILE;
        tile.m = owners > tile.start ? (owners - tile.start < ZD_TILE ? owners - tile.start : ZD_TILE) : 0;
        tile.sorted = c.sorted.data() + (size_t)t * ZD_TILE;
        tile.tmp = c.tmp.data() + (size_t)t * ZD_TILE;
        tile.rank = c.rank.data();
        tile.dir = c.dir.data() + (size_t)t * ZD_DIR_STRIDE;
        tile.dir_prev = nullptr;
        tile.hib = nullptr;
        tile.cnt = nullptr;
        HsLds lds;
        for (int ph = 0; ph < HS_PHASES; ph++)
            for (int w = 0; w < HS_WAVES; w++)
                hash_sort_phase(tile, &lds, w, ph);
    }
    for (uint32_t t = 0; t < c.ntiles; t++) { /* kernel 1b */
        HsTile tile;
        memset(&tile, 0, sizeof tile);
        tile.in = c.in.data();
        tile.n = n;
        tile.start = t * ZD_TILE;
        tile.m = owners > tile.start ? (owners - tile.start < ZD_TILE ? owners - tile.start : ZD_TILE) : 0;
        tile.rank = c.rank.data();
        tile.dir = c.dir.data() + (size_t)t * ZD_DIR_STRIDE;
        tile.dir_prev = t ? c.dir.data() + (size_t)(t - 1) * ZD_DIR_STRIDE : nullptr;
        tile.hib = c.hib.data();
        tile.cnt = c.cnt.data();
        for (int w = 0; w < HS_WAVES; w++)
            hs_link_prev(tile, w);
    }
}

/* kernel 1c (match_table.h), as k_match_table runs it: only with the default window and hash size,
 * levels 4-9, a match-finding strategy; switched with emu_set_table */
static uint32_t g_stair_min = 0; /* (the emulation searches every chain as a staircase unless told otherwise) */
extern "C" void emu_set_stair_min(uint32_t v) { g_stair_min = v; }
int g_sg_one = 1; /* lz_parse_seg.h: chains below the budget searched by all lanes at once (SG_EVAL_ONE) */
extern "C" void emu_set_one(int on) { g_sg_one = on; }
static int g_use_table = 1;
static uint32_t g_mt_cap = MT_CAP;
extern "C" void emu_set_table_cap(uint32_t cap) { g_mt_cap = cap; }
extern "C" void emu_set_table(int on) { g_use_table = on; }
extern "C" { unsigned long long g_mt_cnt[4]; } /* [0] entries, [1] incomplete ones, [2] ones that also answer for longer prev_lengths */
static void build_table(EmuChains &c, int level, int strategy)
{
    c.r2.clear();
    const ZdLevel cfg = level_cfg(level);
    if (!g_use_table || !cfg.slow || cfg.wsize != ZD_TILE || cfg.hbits != 15u || strategy == 2 || strategy == 3)
        return;
    c.r2.assign((size_t)c.n + 64, 0xdeadbeefu);
    MtLds *lds = (MtLds *)malloc(sizeof(MtLds));
    for (uint32_t t = 0; t < c.ntiles && t * ZD_TILE < c.n; t++) {
        memset(lds, 0x3C, sizeof(MtLds));
        MtJob job;
        job.in = c.in.data();
        job.n = c.n;
        job.start = t * ZD_TILE;
        job.sorted = c.sorted.data();
        job.rank = c.rank.data();
        job.hib = c.hib.data();
        job.cnt = c.cnt.data();
        job.r2 = c.r2.data();
        job.cfg = cfg;
        job.strategy = (uint32_t)strategy;
        job.cap = g_mt_cap;
        for (int w = 0; w < MT_WAVES; w++)
            mt_phase_load(job, lds, w);
        const uint32_t base0 = sg_base(job.cfg, job.start, job.n);
        for (int w = 0; w < MT_WAVES; w++)
            mt_phase_search(job, lds, w, base0);
    }
    free(lds);
    for (uint32_t p = 0; p < c.n; p++) {
        g_mt_cnt[0]++;
        g_mt_cnt[1] += (c.r2[p] & MT_INCOMPLETE) != 0;
        g_mt_cnt[2] += (c.r2[p] & MT_RLOK) != 0;
    }
}


/* Every complete entry of the match table against longest_match done the reference's way (the walk
 * along p's own chain with its budget, src/deflate.c:1400-1518): returns the number of entries that
 * differ; *incomplete = entries left to the parser.  An entry flagged MT_RLOK is also checked
 * for the prev_length the lazy parse would ask with (the length in the entry before it). */
static uint32_t emu_chain_entry(const EmuChains &c, uint32_t p, uint32_t v)
{
    const uint32_t cn = c.cnt[p], nA = cn & 0xffffu;
    const uint32_t *run = c.sorted.data() + (size_t)(p >> 15) * ZD_TILE;
    if (v < nA)
        return (p & ~ZD_TILE_MASK) + (run[c.rank[p] - 1u - v] & ZD_TILE_MASK);
    return (p & ~ZD_TILE_MASK) - ZD_TILE + ((run - ZD_TILE)[c.hib[p] - (v - nA)] & ZD_TILE_MASK);
}
static uint32_t emu_ref_longest_match(const EmuChains &c, const ZdLevel &cfg, uint32_t p, uint32_t b0, int strategy)
{
    const uint32_t n = c.n;
    if ((uint64_t)p + 3u > n || b0 >= cfg.lazy)
        return MT_NONE;
    const uint32_t look = n - p, cn = c.cnt[p], total = (cn & 0xffffu) + (cn >> 16);
    if (!total)
        return MT_NONE;
    const uint32_t base = sg_base(cfg, p, n);
    const uint32_t floor_pos = p - base > ZD_MAX_DIST ? p - ZD_MAX_DIST : base;
    const uint32_t q0 = emu_chain_entry(c, p, 0);
    if (!(q0 > base && p - q0 <= ZD_MAX_DIST) || b0 >= look)
        return MT_NONE;
    const uint32_t cap = look < 258u ? look : 258u, nice = cfg.nice < look ? cfg.nice : look;
    uint32_t budget = b0 >= cfg.good ? (uint32_t)cfg.chain >> 2 : cfg.chain;
    const uint8_t *in = c.in.data();
    uint32_t best = b0, where = 0;
    for (uint32_t v = 0; v < total; v++) {
        const uint32_t q = emu_chain_entry(c, p, v);
        if (v && !(q > floor_pos))
            break;
        if (in[q + best] == in[p + best] && in[q + best - 1] == in[p + best - 1] && in[q] == in[p] && in[q + 1] == in[p + 1]) {
            uint32_t len = 2;
            while (len < cap && in[q + len] == in[p + len])
                len++;
            if (len > best) {
                best = len;
                where = q;
                if (len >= nice)
                    break;
            }
            if (--budget == 0)
                break;
        }
    }
    uint32_t len = best < look ? best : look;
    if (len <= 5u) {
        if (strategy == 1)
            len = 2;
        else if (len == 3u && p - where > ZD_TOO_FAR)
            len = 2;
    }
    if (len <= b0 || best == b0)
        return MT_NONE;
    return MT_PACK(len, p - where);
}
extern "C" int emu_table_check(const uint8_t *src, uint32_t n, int level, int strategy, uint32_t *incomplete)
{
    EmuChains c;
    build_chains(c, src, n);
    build_table(c, level, strategy);
    *incomplete = 0;
    if (c.r2.empty())
        return -1;
    const ZdLevel cfg = level_cfg(level);
    int bad = 0;
    for (uint32_t p = 0; p < n; p++) {
        const uint32_t e = c.r2[p];
        if (e & MT_INCOMPLETE) {
            (*incomplete)++;
            continue;
        }
        if ((e & 0xffffffu) != emu_ref_longest_match(c, cfg, p, 2u, strategy))
            bad++;
        if ((e & MT_RLOK) && p && !(c.r2[p - 1] & MT_INCOMPLETE)) {
            const uint32_t key = MT_LEN(c.r2[p - 1]);
            if (key >= 3u && key < cfg.lazy) {
                const uint32_t want = emu_ref_longest_match(c, cfg, p, key, strategy);
                const uint32_t got = MT_LEN(e) > key ? (e & 0xffffffu) : MT_NONE;
                if (want != got)
                    bad++;
            }
        }
    }
    return bad;
}

static int g_fast_global = 1;
extern "C" void emu_set_fast_global(int on) { g_fast_global = on; }
int g_seg_mode = 0; /* 0: runtime's choice, 1: force wave-per-buffer, 2: segmented, segments handed out last first,
                       3: segmented, first first (a parser never finds its successors' traces) */
extern "C" void emu_set_seg_mode(int m) { g_seg_mode = m; }

static bool g_job_seg_ok = true; /* sections: may the segmented parser take the run with its joints? */

/* as the runtime does when every buffer of a batch goes to the segmented parser: no k_link_prev, the
 * parser gets the bucket directories and works hib / cnt out itself (lz_parse_seg.h: sg_link) */
static int g_link_in_parser = 1;
static const uint16_t *g_dir = nullptr;
extern "C" void emu_set_link_in_parser(int on) { g_link_in_parser = on; }

static void run_parse_seg(const LzJob &job00, int order)
{
    LzJob job0 = job00;
    if (g_link_in_parser && g_dir) {
        job0.dir = g_dir;
        job0.hib = nullptr; /* (must not be read) */
        job0.cnt = nullptr;
    }
    LzJob job = job0;
    SgLds *lds = (SgLds *)malloc(sizeof(SgLds));
    memset(lds, 0x6B, sizeof(SgLds));
    std::vector<uint32_t> tok((size_t)SG_NS * SG_TOKCAP, 0xDDDDDDDD);
    std::vector<uint16_t> sidx((size_t)SG_NS * SG_TRACE, 0xDDDD);
    SgScratch scr = {tok.data(), sidx.data()};
    for (int w = 0; w < SG_W; w++)
        sg_init(lds, w);
    lds->emu_ascending = order == 3;
    /* the phases of a run with joints, as k_parse_seg goes through them */
    uint32_t si = 0, nph = job0.nsched ? job0.n0 : job0.n;
    for (;;) {
        nph = sg_phase_end(job0, nph, &si);
        const bool goes_on = si < job0.nsched;
        job.n = nph;
        job.more = goes_on ? 1u : job0.more;
        while (!lds->finished) {
            for (int w = 0; w < SG_W; w++)
                sg_phase_begin(job, lds, w);
            /* waves run one after the other here, so the first one drains the queue */
            do {
                for (int w = SG_W - 1; w >= 0; w--)
                    if (job.r2 != nullptr)
                        sg_phase_parse<true>(job, lds, scr, w);
                    else
                        sg_phase_parse<false>(job, lds, scr, w);
                for (int w = 0; w < SG_W; w++)
                    sg_phase_resolve(job, lds, scr, w);
            } while (lds->redo);
        }
        if (!goes_on)
            break;
        nph = job0.sched[si].new_n;
        si++;
        for (int w = 0; w < SG_W; w++)
            sg_next_phase(lds, w, job.n);
    }
    free(lds);
}

static void run_parse(const LzJob &job)
{
    if (job.strategy == 2u || job.strategy == 3u) { /* Z_HUFFMAN_ONLY, Z_RLE */
        SpLds *lds = (SpLds *)malloc(sizeof(SpLds));
        memset(lds, 0x5D, sizeof(SpLds));
        if (job.nsched)
            lz_parse_simple_joints(job, lds);
        else if (job.strategy == 2u)
            lz_parse_huff(job, lds);
        else
            lz_parse_rle(job, lds);
        free(lds);
        return;
    }
    if (job.cfg.slow && (job.nsched == 0 || g_job_seg_ok) &&
        (g_seg_mode >= 2 || (g_seg_mode == 0 && job.n > 18432u))) {
        run_parse_seg(job, g_seg_mode == 3 ? 3 : 2);
        return;
    }
    if (job.cfg.slow) {
        /* the ring class the runtime would pick for this length; runs with joints keep a hole map */
#define EMU_LAZY(LT)                                 \
    do {                                             \
        LT *lds = (LT *)malloc(sizeof(LT));          \
        memset(lds, 0xA5, sizeof(LT));               \
        lz_parse_lazy<LT>(job, lds);                 \
        free(lds);                                   \
    } while (0)
        if (job.nsched) {
            if (job.n > 18432u)
                EMU_LAZY(LzLdsJ);
            else if (job.n > 10240u)
                EMU_LAZY(LzLdsJ16k);
            else if (job.n > 6144u)
                EMU_LAZY(LzLdsJ8k);
            else
                EMU_LAZY(LzLdsJ4k);
        } else if (job.n > 18432u)
            EMU_LAZY(LzLds);
        else if (job.n > 10240u)
            EMU_LAZY(LzLds16k);
        else if (job.n > 6144u)
            EMU_LAZY(LzLds8k);
        else
            EMU_LAZY(LzLds4k);
#undef EMU_LAZY
    } else {
        if (g_fast_global) { /* the product's choice: the window read from the input, no LDS ring */
            LzLdsFastG *lds = (LzLdsFastG *)malloc(sizeof(LzLdsFastG));
            memset(lds, 0xA5, sizeof(LzLdsFastG));
            lz_parse_greedy<LzLdsFastG>(job, lds);
            free(lds);
        } else {
            LzLdsFast *lds = (LzLdsFast *)malloc(sizeof(LzLdsFast));
            memset(lds, 0xA5, sizeof(LzLdsFast));
            lz_parse_greedy<LzLdsFast>(job, lds);
            free(lds);
        }
    }
}

extern "C" int emu_sort(const uint8_t *src, uint32_t n, uint32_t *sorted_out, uint16_t *rank_out,
                        uint16_t *dir_out)
{
    EmuChains c;
    build_chains(c, src, n);
    memcpy(sorted_out, c.sorted.data(), c.sorted.size() * 4);
    memcpy(rank_out, c.rank.data(), (size_t)n * 2);
    memcpy(dir_out, c.dir.data(), c.dir.size() * 2);
    return (int)c.ntiles;
}

extern "C" int emu_parse(const uint8_t *src, uint32_t n, int level, int strategy, uint32_t *syms,
                         uint32_t *nsyms, ZdBlockRec *blocks, uint32_t *nblocks)
{
    if (level < 1 || level > 9)
        return -2;
    EmuChains c;
    build_chains(c, src, n);
    build_table(c, level, strategy);
    LzJob job;
    ZdParseOut out = {0, 0};
    job.in = c.in.data();
    job.n = n;
    job.sorted = c.sorted.data();
    job.rank = c.rank.data();
    job.hib = c.hib.data();
    job.cnt = c.cnt.data();
    job.dir = nullptr;
    g_dir = c.dir.data();
    job.r2 = c.r2.empty() ? nullptr : c.r2.data();
    job.stair_min = g_stair_min;
    job.syms = syms;
    job.blocks = blocks;
    job.out = &out;
    job.cfg = level_cfg(level);
    job.strategy = (uint32_t)strategy;
    job.more = 0;
    job.sched = nullptr;
    job.nsched = 0;
    job.n0 = n;
    job.ntot = n;
    run_parse(job);
    *nsyms = out.nsyms;
    *nblocks = out.nblocks;
    return 0;
}

extern "C" uint32_t emu_adler32(const uint8_t *src, uint32_t n)
{
    std::vector<uint8_t> in((size_t)n + 64, 0);
    memcpy(in.data(), src, n);
    return ck_adler32(in.data(), n);
}

extern "C" uint32_t emu_crc32(const uint8_t *src, uint32_t n)
{
    std::vector<uint8_t> in((size_t)n + 64, 0);
    memcpy(in.data(), src, n);
    CkLds lds;
    return ck_crc32(in.data(), n, &lds);
}

/* the whole deflate pipeline, kernel by kernel, for one buffer */
extern "C" int emu_compress(const uint8_t *src, uint32_t n, int level, int wrap, int strategy,
                            uint8_t *out, uint32_t out_cap, uint32_t *out_len)
{
    if (level < 1 || level > 9)
        return -2;
    EmuChains c;
    build_chains(c, src, n);
    build_table(c, level, strategy);
    std::vector<uint32_t> syms((size_t)n + 64);
    uint32_t max_blocks = n / ((1u << (g_mem_level + 6)) - 1u) + 2;
    std::vector<ZdBlockRec> recs(max_blocks);
    std::vector<ZdBlockPlan> plans(max_blocks);
    ZdParseOut po = {0, 0};
    LzJob job;
    job.in = c.in.data();
    job.n = n;
    job.sorted = c.sorted.data();
    job.rank = c.rank.data();
    job.hib = c.hib.data();
    job.cnt = c.cnt.data();
    job.dir = nullptr;
    g_dir = c.dir.data();
    job.r2 = c.r2.empty() ? nullptr : c.r2.data();
    job.stair_min = g_stair_min;
    job.syms = syms.data();
    job.blocks = recs.data();
    job.out = &po;
    job.cfg = level_cfg(level);
    job.strategy = (uint32_t)strategy;
    job.more = 0;
    job.sched = nullptr;
    job.nsched = 0;
    job.n0 = n;
    job.ntot = n;
    run_parse(job);

    ZdBuf buf;
    memset(&buf, 0, sizeof buf);
    buf.in_len = n;
    buf.max_blocks = max_blocks;
    buf.out_cap = out_cap;
    buf.level = (uint32_t)level;
    buf.wrap = (uint32_t)wrap;
    buf.strategy = (uint32_t)strategy;
    buf.wbits = (uint32_t)g_wbits;
    ZdResult res;
    memset(&res, 0, sizeof res);
    CkLds ck;
    res.adler = wrap == 1 ? ck_adler32(c.in.data(), n) : wrap == 2 ? ck_crc32(c.in.data(), n, &ck) : 0;

    for (uint32_t b = 0; b < po.nblocks; b++) {
        HpLds hl;
        memset(&hl, 0x5A, sizeof hl);
        huff_plan_block(syms.data() + recs[b].sym_begin, &recs[b], (uint32_t)strategy, &plans[b], &hl);
    }
    std::vector<uint32_t> outw(((size_t)out_cap + 64) / 4 + 4, 0xCDCDCDCD);
    layout_buffer(&buf, &po, recs.data(), plans.data(), &res, (uint8_t *)outw.data());
    for (uint32_t b = 0; b < po.nblocks; b++) {
        BeLds bl;
        memset(&bl, 0x77, sizeof bl);
        emit_block(c.in.data(), syms.data() + recs[b].sym_begin, &recs[b], &plans[b], outw.data(), &bl);
    }
    *out_len = res.out_len;
    if (res.status == 0)
        memcpy(out, outw.data(), res.out_len);
    return res.status;
}

extern "C" int emu_uncompress(const uint8_t *src, uint32_t n, int window_bits, uint8_t *dst, uint32_t cap,
                              uint32_t *out_len, uint32_t *consumed)
{
    std::vector<uint8_t> in((size_t)n + 64, 0);
    memcpy(in.data(), src, n);
    std::vector<uint8_t> out((size_t)cap + 64, 0xEE);
    InfJob job = {in.data(), n, out.data(), cap, window_bits};
    InfLds *lds = (InfLds *)malloc(sizeof(InfLds));
    memset(lds, 0x3C, sizeof(InfLds));
    static uint32_t crc_table[1][256];
    lds->cktab = crc_table;
    InfResult res = {};
    inflate_with_resync(job, lds, &res);
    free(lds);
    *out_len = res.out_len;
    *consumed = res.consumed;
    memcpy(dst, out.data(), res.out_len <= cap ? res.out_len : cap);
    return res.status;
}


/* ---- zsc_compress with source_len > max_block_len (sections.h), kernel by kernel ---- */

struct EmuSecRunner {
    const uint8_t *src;
    int level, strategy;
    std::vector<std::vector<std::vector<uint32_t>>> outs; /* [round][job] */
    uint32_t parses = 0;

    int operator()(std::vector<SecRun *> &jobs, uint32_t round)
    {
        outs.resize(round + 1);
        outs[round].resize(jobs.size());
        for (size_t j = 0; j < jobs.size(); j++) {
            SecRun &r = *jobs[j];
            parses++;
            EmuChains c;
            build_chains(c, src + r.start, r.n);
            std::vector<uint32_t> syms((size_t)r.n + 64);
            const uint32_t max_blocks = r.n / ((1u << (g_mem_level + 6)) - 1u) + 2 + (uint32_t)r.sched.size();
            std::vector<ZdBlockRec> recs(max_blocks);
            std::vector<ZdBlockPlan> plans(max_blocks);
            ZdParseOut po = {0, 0};
            LzJob job;
            job.in = c.in.data();
            job.n = r.n;
            job.sorted = c.sorted.data();
            job.rank = c.rank.data();
            job.hib = c.hib.data();
            job.cnt = c.cnt.data();
            job.dir = nullptr;
            g_dir = c.dir.data();
            if (r.sched.empty())
                build_table(c, level, strategy);
            job.r2 = c.r2.empty() ? nullptr : c.r2.data();
            job.stair_min = g_stair_min;
    job.stair_min = g_stair_min;
                    job.syms = syms.data();
            job.blocks = recs.data();
            job.out = &po;
            job.cfg = level_cfg(level);
            job.strategy = (uint32_t)strategy;
            job.more = r.more ? 1u : 0u;
            job.sched = r.sched.data();
            job.nsched = (uint32_t)r.sched.size();
            job.n0 = r.n0;
            job.ntot = r.n;
            g_job_seg_ok = sec_seg_ok(r);
            run_parse(job);
            if (po.nblocks > max_blocks) {
                if (getenv("ZSC_EMU_DEBUG"))
                    fprintf(stderr, "emu: run at %u n %u: %u blocks > %u\n", r.start, r.n, po.nblocks, max_blocks);
                return -2;
            }

            ZdBuf buf;
            memset(&buf, 0, sizeof buf);
            buf.in_len = r.n;
            buf.max_blocks = max_blocks;
            buf.out_cap = r.n + (r.n >> 3) + 1024u;
            buf.level = (uint32_t)level;
            buf.wrap = 0;
            buf.strategy = (uint32_t)strategy;
            buf.wbits = (uint32_t)g_wbits;
            buf.more = job.more;
            ZdResult res;
            memset(&res, 0, sizeof res);
            for (uint32_t b = 0; b < po.nblocks; b++) {
                HpLds hl;
                memset(&hl, 0x5A, sizeof hl);
                huff_plan_block(syms.data() + recs[b].sym_begin, &recs[b], (uint32_t)strategy, &plans[b], &hl);
            }
            std::vector<uint32_t> &outw = outs[round][j];
            outw.assign(((size_t)buf.out_cap + 64) / 4 + 4, 0xCDCDCDCD);
            layout_buffer(&buf, &po, recs.data(), plans.data(), &res, (uint8_t *)outw.data());
            if (res.status != 0) {
                if (getenv("ZSC_EMU_DEBUG"))
                    fprintf(stderr, "emu: run at %u n %u: layout status %d (out_len %u cap %u)\n", r.start, r.n, res.status, res.out_len, buf.out_cap);
                return res.status;
            }
            for (uint32_t b = 0; b < po.nblocks; b++) {
                BeLds bl;
                memset(&bl, 0x77, sizeof bl);
                emit_block(c.in.data(), syms.data() + recs[b].sym_begin, &recs[b], &plans[b], outw.data(), &bl);
            }
            r.blocks.clear();
            for (uint32_t b = 0; b < po.nblocks; b++) {
                SecBlock sb;
                sb.upto = recs[b].in_begin + recs[b].in_len;
                sb.end_bit = b + 1 < po.nblocks ? plans[b + 1].bit_off : res.bits;
                sb.wend = recs[b].wend;
                sb.at = recs[b].at;
                sb.cut = recs[b].cut;
                sb.last = recs[b].last;
                r.blocks.push_back(sb);
            }
            r.round = round;
            r.job = (uint32_t)j;
        }
        return 0;
    }
};

static uint32_t g_gz_hdr_len = 0; /* emu_set_gz_header_len: a caller's gzip header of this length (0: the plain one) */
extern "C" void emu_set_gz_header_len(uint32_t n) { g_gz_hdr_len = n; }

/* returns the call's ZlibReturn; *rounds_out / *parses_out tell how much work it took */
extern "C" int emu_compress_sections(const uint8_t *src, uint32_t n, uint32_t max_block_len, int level,
                                     int wrap, int strategy, uint8_t *out, uint32_t dest_cap,
                                     uint32_t *out_len, uint32_t *parses_out)
{
    std::vector<SecStream> streams(1);
    SecStream &s = streams[0];
    s.source_len = n;
    s.max_block_len = max_block_len;
    s.dest_cap = dest_cap;
    s.wrap = wrap;
    s.hdr_len = wrap == 1 ? 2u : wrap == 2 ? (g_gz_hdr_len ? g_gz_hdr_len : 10u) : 0u;
    s.need = strategy == 2 ? 1u : strategy == 3 ? ZD_MAX_MATCH + 1u : ZD_MIN_LOOKAHEAD;
    EmuSecRunner runner;
    runner.src = src;
    runner.level = level;
    runner.strategy = strategy;
    const int rc = sec_compress(streams, runner);
    *parses_out = runner.parses;
    *out_len = 0;
    if (getenv("ZSC_EMU_DEBUG"))
        fprintf(stderr, "emu: sec_compress rc %d, stream status %d done %d (sections.h:%d)\n", rc, s.status, (int)s.done, s.broken_line);
    if (rc != 0)
        return rc;
    std::vector<uint8_t> whole((size_t)s.produced + 16, 0xEE);
    for (const SecPiece &pc : s.pieces) {
        uint8_t *o = whole.data() + pc.dst;
        if (pc.kind == SEC_PIECE_RUN) {
            memcpy(o, runner.outs[pc.round][pc.job].data(), pc.len);
        } else if (pc.kind == SEC_PIECE_TAIL) {
            o[0] = ((const uint8_t *)runner.outs[pc.round][pc.job].data())[pc.src] & (uint8_t)pc.mask;
            if (pc.len > 1)
                o[1] = 0;
        } else if (pc.kind == SEC_PIECE_MARKER) {
            o[0] = o[1] = 0;
            o[2] = o[3] = 0xff;
        } else if (pc.kind == SEC_PIECE_HEADER) {
            /* what layout_buffer writes for the wrapper, from a run of nothing */
            ZdBuf buf;
            memset(&buf, 0, sizeof buf);
            buf.out_cap = 64;
            buf.level = (uint32_t)level;
            buf.wrap = (uint32_t)wrap;
            buf.strategy = (uint32_t)strategy;
            buf.wbits = (uint32_t)g_wbits;
            buf.max_blocks = 1;
            ZdParseOut po = {0, 0};
            ZdResult res;
            memset(&res, 0, sizeof res);
            uint32_t tmp[32];
            layout_buffer(&buf, &po, nullptr, nullptr, &res, (uint8_t *)tmp);
            if (wrap == 2 && g_gz_hdr_len)
                memset(o, 0xAA, pc.len); /* zsc_api.c writes the caller's header there */
            else
                memcpy(o, tmp, pc.len);
        } else {
            CkLds ck;
            std::vector<uint8_t> in((size_t)n + 64, 0);
            memcpy(in.data(), src, n);
            const uint32_t c = wrap == 1 ? ck_adler32(in.data(), n) : ck_crc32(in.data(), n, &ck);
            if (wrap == 1) {
                o[0] = (uint8_t)(c >> 24), o[1] = (uint8_t)(c >> 16), o[2] = (uint8_t)(c >> 8), o[3] = (uint8_t)c;
            } else {
                for (int k = 0; k < 4; k++)
                    o[k] = (uint8_t)(c >> (8 * k)), o[4 + k] = (uint8_t)(n >> (8 * k));
            }
        }
    }
    memcpy(out, whole.data(), s.delivered);
    *out_len = s.delivered;
    return s.status;
}
