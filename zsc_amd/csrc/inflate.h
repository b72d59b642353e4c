/*
 * inflate.h -- kernel 5: DEFLATE decoding, one GROUP of lanes per stream (16 lanes on the GPU:
 * four streams share a wavefront; wave_group.h).
 *
 * Restates the reference's inflate() state machine for one-shot use
 * (src/inflate.c:704-1404: header :740-954, block type :975-1009, stored :1010-1049,
 * dynamic header :1050-1178, symbols :1182-1321, trailers :1322-1354), its code
 * validation (inflate_table, src/inftrees.c:130-177) and the copy semantics of
 * inflate_fast (src/inffast.c:125-297), as the one-shot wrapper drives them
 * (src/zsc_uncompr.c:44-154).
 *
 * Decoding one stream is bit-serial, so the parallelism is across streams (config
 * 4 has a million of them) plus, inside the group:
 *   - Huffman decoding without lookup tables: the next 15 bits are bit-reversed so
 *     the code sits MSB-first, lane l tests "is the code one of the length-l codes"
 *     (code_l - first[l] < count[l]); the lowest set bit of the group's ballot slice is
 *     the code length.  Decoder state is 1.7 KiB of LDS per stream (not zlib's 5.8 KiB
 *     of tables): 20 waves = 80 streams per CU;
 *   - match copies are lane-parallel even when source and destination overlap:
 *     byte i of a copy comes from pos - dist + (i mod dist), which always lies
 *     before the copy's first byte;
 *   - input is consumed from a 64-byte chunk held across the lanes of one VGPR
 *     (ds_bpermute); every output byte is stored to dst as it is made and kept in a
 *     512-byte LDS ring, from which near matches are copied (farther ones are read
 *     back from dst).
 * Group-uniform values live in vector registers, so the four groups of a wave share the
 * instruction stream where their paths agree; the symbol loop is shaped for that (one exit,
 * literal runs as an inner loop; see the comment there and DESIGN.md 5b).
 * Byte counts follow the reference: bytes are "consumed" once pulled into the bit
 * buffer, which for both its fast and slow paths is ceil(bits used / 8).
 */
#ifndef ZSC_INFLATE_H
#define ZSC_INFLATE_H

#include "checksum.h"
#include "wave.h"

#ifndef INF_STAGE
#define INF_STAGE 512u
#endif
static_assert(INF_STAGE >= 512u && (INF_STAGE & (INF_STAGE - 1u)) == 0, "the ring is a power of two and holds the longest match");

template <int NSYM>
struct InfCodeT {
    uint16_t count[16];
    uint16_t first[16]; /* first canonical code of each length */
    uint16_t offs[16];  /* index of that code's symbol in sym[] */
    uint16_t fill[16];  /* scratch of inf_build */
    uint16_t sym[NSYM];
    uint32_t max_len;
    uint32_t empty;
};

/* per stream: 1.7 KiB with a 512-byte ring, so that four streams per wave (+ the CRC table) leave
 * room for twenty waves on a CU (measured on BASELINE config 4: 12 waves 30.4 GB/s, 16 waves 40.9,
 * 20 waves 43.2 -- by then the vector ALUs are the limit).  The code-length code is dead once the
 * lengths are read, before the distance code is built, and shares its place; the CRC routine's
 * exchange area lies in the ring, which is not needed whenever a check value is computed (gzip
 * header: nothing decoded yet; trailer: nothing left to decode). */
typedef struct {
    InfCodeT<288> lit;
    union {
        InfCodeT<32> dist;
        InfCodeT<20> cl;
    };
    uint8_t lens[320];
    __attribute__((aligned(16))) uint8_t stage[INF_STAGE + 8]; /* the last INF_STAGE bytes of output, indexed modulo INF_STAGE: the source of near matches */
    uint32_t (*cktab)[256];         /* the CRC byte-loop table (1 KiB), shared by the streams of a wave */
} InfLds;
#define INF_CKX(lds) ((uint32_t *)(lds)->stage) /* 64 words */

/* where decoding goes on after an inflateSync: the stream is inflated again from there */
typedef struct {
    uint32_t state;  /* 0 not started, 1 a data error happened: resynchronise and go on, 2 finished */
    uint32_t out_pos, errors, gzip;
    uint32_t sy_lo, sy_hi, sy_rb; /* what the reference's bit buffer held at the error */
} InfResume;

typedef struct {
    const uint8_t *src;
    uint32_t n;
    uint8_t *dst;
    uint32_t cap;
    int32_t window_bits;
} InfJob;

typedef struct {
    int32_t status; /* ZlibReturn of the reference's zsc_uncompress2 */
    uint32_t out_len;
    uint32_t consumed;
    uint32_t pad;
} InfResult;

/* wave-uniform bit reader over a lane-distributed 256-byte chunk */
typedef struct {
    uint64_t hold;
    uint32_t bits;     /* valid bits in hold */
    uint32_t next;     /* next input byte to pull into hold */
    uint32_t chunk_at; /* input offset of the chunk held in `cur` (multiple of 256) */
} InfBits;

/* bits consumed so far: every byte pulled into hold, less what is still there */
#define BR_USED ((uint64_t)br.next * 8u - br.bits)

/* bytes a .. a+3 of the stream as a dword, zero where the stream has ended.  On the GPU the last,
 * partial dword is read whole and masked: streams start on 16-byte boundaries and 64 readable bytes
 * follow the last one (include/zsc_hip.h), so the read stays inside mapped memory; the host
 * emulation reads exactly the stream's bytes. */
DEV uint32_t inf_input_dword(const uint8_t *src, uint32_t a, uint32_t n)
{
#ifdef ZSC_WAVE_EMU
    uint32_t v = 0;
    if (a + 4 <= n)
        v = ld_u32(src + a);
    else
        for (uint32_t j = 0; j < 4; j++)
            if (a + j < n)
                v |= (uint32_t)src[a + j] << (8 * j);
    return v;
#else
    if (a >= n)
        return 0;
    const uint32_t v = ld_u32(src + a);
    return n - a >= 4u ? v : v & ((1u << (8u * (n - a))) - 1u);
#endif
}

#define INF_OK 0
#define INF_END 1
#define INF_NEED_DICT 2
#define INF_DATA (-3)
#define INF_BUF (-5)

/* The decoder is written for a group of lanes (wave_group.h): INF_GROUP lanes per stream,
 * 64 / INF_GROUP streams per wavefront.  A stream is decoded by serial control flow; with a
 * whole wave per stream that control flow is scalar instructions for one stream (47 of them
 * per output byte, against 13 vector ones: profiles/), with four streams per wave the same
 * instructions, now vector ones, serve four.  The groups stay close together in the symbol
 * loop, which is where the time goes. */
#ifndef INF_GROUP
#define INF_GROUP 16
#endif
#undef ZSC_GROUP
#define ZSC_GROUP INF_GROUP
#include "wave_group.h"
#if !defined(ZSC_WAVE_EMU) && INF_GROUP != 64
#define CK_FN(name) ckg_##name
#include "checksum_impl.h"
#undef CK_FN
#define INF_CK(name) ckg_##name
#else
#define INF_CK(name) ck_##name
#endif
#define INF_CHUNK (4u * GRP) /* input bytes held across the lanes of the group */
#define INF_SLOTS (GRP >= 16 ? 1 : 2) /* code lengths a lane tests: l, and l + GRP in a group of 8 */

/* build a canonical decoder from code lengths.  kind 0: code-length code, 1: literal/
 * length, 2: distance.  Returns 0, or -1 for an invalid set (src/inftrees.c:168-177). */
template <class CT>
DEV int inf_build(CT *c, const uint8_t *lens, int n, int kind)
{
    int rc = 0;
    ON_GLANE0
    {
        for (int i = 0; i < 16; i++)
            c->count[i] = 0;
        for (int i = 0; i < n; i++)
            c->count[lens[i]]++;
        int max = 15;
        while (max >= 1 && c->count[max] == 0)
            max--;
        c->max_len = (uint32_t)max;
        c->empty = max == 0;
        if (max != 0) {
            int left = 1;
            for (int l = 1; l <= 15; l++) {
                left <<= 1;
                left -= c->count[l];
                if (left < 0) {
                    rc = -1;
                    break;
                }
            }
            if (rc == 0 && left > 0 && (kind == 0 || max != 1))
                rc = -1;
            if (rc == 0) {
                uint32_t code = 0, idx = 0;
                for (int l = 1; l <= 15; l++) {
                    code <<= 1;
                    c->first[l] = (uint16_t)code;
                    c->offs[l] = (uint16_t)idx;
                    code += c->count[l];
                    idx += c->count[l];
                }
                for (int l = 0; l < 16; l++)
                    c->fill[l] = c->offs[l];
                for (int i = 0; i < n; i++)
                    if (lens[i])
                        c->sym[c->fill[lens[i]]++] = (uint16_t)i;
            }
        }
        c->count[0] = (uint16_t)(rc + 1); /* publish rc to the other lanes: 1 ok, 0 bad */
    }
    WAVE_SYNC();
    return (int)c->count[0] - 1;
}

/* inflateSync's search (reference src/inflate.c:1523-1604) for the whole wave: the first
 * 00 00 FF FF in what the reference's bit buffer holds at the error (sy_rb bits of the stream
 * from bit sy_start, after its "hold <<= bits & 7") followed by the input from byte nin on.
 * Returns the number of input bytes up to the end of the pattern, or 0xffffffff when there
 * is none. */
DEV uint32_t inf_sync_search(const uint8_t *src, uint32_t n, uint64_t sy_start, uint32_t sy_rb,
                                      uint32_t nin)
{
    uint32_t hold = 0;
    for (uint32_t k = 0; k < sy_rb; k += 8u) {
        const uint64_t bit = sy_start + k;
        const uint32_t by = (uint32_t)(bit >> 3), sh = (uint32_t)(bit & 7u);
        uint32_t two = GUNI(src[by]);
        if (sh && by + 1u < n)
            two |= GUNI(src[by + 1u]) << 8;
        hold |= ((two >> sh) & 0xffu) << k;
    }
    if (sy_rb < 32u)
        hold &= (1u << sy_rb) - 1u;
    uint32_t rb = sy_rb;
    hold <<= rb & 7u; /* sic, :1571 */
    rb -= rb & 7u;
    const uint32_t nh = rb >> 3; /* bytes of hold searched before the input */
    const uint32_t vlen = nh + (n - nin);
    for (uint32_t base = 0; base + 4u <= vlen; base += GRP) {
        LANEVAR(int, _hit);
        FOR_GLANES
        {
            const uint32_t m = base + (uint32_t)GLANE;
            int ok = m + 4u <= vlen;
            for (uint32_t j = 0; j < 4u && ok; j++) {
                const uint32_t i = m + j;
                const uint32_t c = i < nh ? (hold >> (8u * i)) & 0xffu : src[nin + (i - nh)];
                ok = c == (j < 2u ? 0u : 0xffu);
            }
            LV(_hit) = ok;
        }
        const uint64_t hm = GBALLOT(_hit);
        if (hm != 0) {
            const uint32_t found = base + (uint32_t)CTZ64(hm);
            return found + 4u > nh ? found + 4u - nh : 0u;
        }
    }
    return 0xffffffffu;
}

/* the whole stream; mirrors zsc_uncompress_gzip2 with gz_head == NULL */
/* One inflate() call of zsc_uncompress's loop (reference src/zsc_uncompr.c:104-125): decodes
 * until the stream ends or fails.  Returns 1 after a data error: *rs then holds what
 * inflateSync needs, and the stream is entered once more -- inflateSync's search for the
 * next flush marker first, then decoding from there -- by another launch of the kernel, not
 * by a loop in here: wrapping the decoder in an outer loop doubles its register count (135
 * instead of 68 VGPRs, occupancy 3 instead of 7), and even the search alone, placed after the
 * decode loop, costs it a wave per SIMD; the sound streams would pay for the damaged ones. */
DEV int inflate_stream(const InfJob &job, InfLds *lds, InfResult *res, InfResume *rs)
{
    const uint8_t *src = job.src;
    const uint32_t n = job.n, cap = job.cap;
    uint8_t *dst = job.dst;

    /* inflateReset2, reference src/inflate.c:341-356 */
    int wrap, wb = job.window_bits;
    if (wb < 0) {
        wrap = 0;
        wb = -wb;
    } else {
        wrap = (wb >> 4) + 5;
        if (wb < 48)
            wb &= 15;
    }

    InfBits br;
    br.hold = 0;
    br.bits = 0;
    br.next = 0;
    br.chunk_at = 0;
    LANEVAR(uint32_t, cur); /* dword GLANE of the current 256-byte input chunk */
    FOR_GLANES { LV(cur) = inf_input_dword(src, 4u * (uint32_t)GLANE, n); }

    const int resumed = GUNI(rs->state) == 1u;
    uint32_t pos = 0; /* output bytes produced (each is stored to dst as it is made) */
    uint32_t dmax = 32768u;
    int gzip = 0;
    int rc = INF_OK;
    int exhausted = 0; /* ran out of input */
    uint32_t data_errors = 0; /* data errors survived by resynchronising (zsc_uncompress's loop) */
    uint32_t out_base = 0;    /* output of the current inflate() call starts here */
    uint64_t sy_start = 0;    /* at a data error: where the reference's bit buffer starts (bit offset) */
    uint32_t sy_rb = 0;       /* ... and how many bits it holds */
    uint32_t fail_line = 0; /* source line of the check that rejected the stream (diagnostics) */

/* top up the bit buffer to at least 32 bits (or to the end of the input) */
#define INF_REFILL()                                                                          \
    do {                                                                                      \
        while (br.bits <= 32 && br.next < n) {                                                \
            if (br.next - br.chunk_at >= INF_CHUNK) {                                              \
                br.chunk_at += INF_CHUNK;                                                          \
                FOR_GLANES                                                                     \
                {                                                                             \
                    LV(cur) = inf_input_dword(src, br.chunk_at + 4u * (uint32_t)GLANE, n);     \
                }                                                                             \
            }                                                                                 \
            const uint32_t _o = br.next - br.chunk_at;                                        \
            const uint32_t _w = GREADLANE(cur, _o >> 2);                                       \
            if ((_o & 3u) == 0 && n - br.next >= 4u) { /* (br.bits <= 32) the whole dword */    \
                br.hold |= (uint64_t)_w << br.bits;                                           \
                br.bits += 32u;                                                               \
                br.next += 4u;                                                                \
                continue;                                                                     \
            }                                                                                 \
            /* take the bytes of this dword from the current one on (up to 4) */              \
            uint32_t _take = 4u - (_o & 3u);                                                  \
            if (_take > n - br.next)                                                          \
                _take = n - br.next;                                                          \
            if (_take * 8u > 64u - br.bits)                                                   \
                _take = (64u - br.bits) >> 3;                                                 \
            if (_take == 0)                                                                   \
                break;                                                                        \
            uint64_t _v = (uint64_t)(_w >> (8u * (_o & 3u)));                                 \
            if (_take < 4)                                                                    \
                _v &= (1ull << (8u * _take)) - 1ull;                                          \
            br.hold |= _v << br.bits;                                                         \
            br.bits += 8u * _take;                                                            \
            br.next += _take;                                                                 \
        }                                                                                     \
    } while (0)

/* what happens when the input runs out / the output is full: by default straight to the exit
 * bookkeeping; the symbol loop redefines both (and INF_BAD) to leave through ONE exit with an
 * event code, so that the loop keeps a simple shape (the compiler otherwise threads a dispatch
 * variable and copies of the live state through every join of the loop) */
#define INF_ON_EXHAUST   \
    do {                 \
        exhausted = 1;   \
        rc = INF_BUF;    \
        goto done;       \
    } while (0)
#define INF_ON_FULL      \
    do {                 \
        rc = INF_BUF;    \
        goto done;       \
    } while (0)

/* need nb (<= 32) bits; on failure the input is exhausted */
#define INF_NEED(nb)                     \
    do {                                 \
        const uint32_t _need = (uint32_t)(nb); \
        if (br.bits < _need) {           \
            INF_REFILL();                \
            if (br.bits < _need)         \
                INF_ON_EXHAUST;          \
        }                                \
    } while (0)

#define INF_TAKE(var, nb)                                          \
    do {                                                           \
        const uint32_t _nb = (uint32_t)(nb);                       \
        (var) = (uint32_t)(br.hold & ((1ull << _nb) - 1ull));      \
        br.hold >>= _nb;                                           \
        br.bits -= _nb;                                            \
    } while (0)

/* A data error.  inflateSync starts its search in the bits the reference has buffered at
 * that point (src/inflate.c:1570-1582), so every error site states them: BACK bits already
 * taken here that the reference has not dropped yet, RB bits in its `hold` in all.  The
 * usual case is "whatever is left of the current byte". */
#define INF_BADX(BACK, RB)                                  \
    do {                                                    \
        sy_start = BR_USED - (uint64_t)(int64_t)(BACK);     \
        sy_rb = (uint32_t)(RB);                             \
        fail_line = __LINE__;                               \
        goto bad;                                           \
    } while (0)
#define INF_BAD INF_BADX(0, (8u - (uint32_t)(BR_USED & 7u)) & 7u)

/* put the bit reader at byte P of the input */
#define INF_SEEK(P)                                                                           \
    do {                                                                                      \
        br.hold = 0;                                                                          \
        br.bits = 0;                                                                          \
        br.next = (P);                                                                        \
        if (br.next - br.chunk_at >= INF_CHUNK || br.next < br.chunk_at) {                         \
            br.chunk_at = br.next & ~(INF_CHUNK - 1u);                                                    \
            FOR_GLANES                                                                         \
            {                                                                                 \
                LV(cur) = inf_input_dword(src, br.chunk_at + 4u * (uint32_t)GLANE, n);         \
            }                                                                                 \
        }                                                                                     \
    } while (0)


/* decode one symbol of code C into `sym`: -2 when the bits are not a code of the set */
#define INF_DECODE(C, OUTSYM)                                                                    \
    do {                                                                                      \
        if (br.bits < 15)                                                                     \
            INF_REFILL();                                                                     \
        if ((C)->empty) {                                                                     \
            /* a table of invalid-code markers of length 1 (src/inftrees.c:150-158) */        \
            INF_NEED(1);                                                                      \
            uint32_t _d1;                                                                     \
            INF_TAKE(_d1, 1); /* DROPBITS(here.bits) precedes the op test, :1217-1236 */      \
            (void)_d1;                                                                        \
            (OUTSYM) = -2;                                                                       \
            break;                                                                            \
        }                                                                                     \
        const uint32_t _peek = (uint32_t)br.hold & 0x7fffu;                                   \
        const uint32_t _r = BREV32(_peek) >> 17; /* the 15 bits MSB first */                  \
        LANEVAR(int, _hit);                                                                   \
        LANEVAR(int, _hit2);                                                                  \
        FOR_GLANES                                                                             \
        {                                                                                     \
            /* lane l tests the code lengths l and (in a group of fewer than 16 lanes) l + GRP */ \
            int _ok[2] = {0, 0};                                                              \
            for (int _k = 0; _k < INF_SLOTS; _k++) {                                          \
                const int _l = GLANE + _k * (int)GRP;                                         \
                if (_l >= 1 && _l <= (int)(C)->max_len) {                                     \
                    uint32_t _c = _r >> (15 - _l);                                            \
                    _ok[_k] = (uint32_t)(_c - (C)->first[_l]) < (uint32_t)(C)->count[_l];     \
                }                                                                             \
            }                                                                                 \
            LV(_hit) = _ok[0];                                                                \
            LV(_hit2) = _ok[1];                                                               \
        }                                                                                     \
        const uint64_t _m = GBALLOT(_hit) | (INF_SLOTS > 1 ? GBALLOT(_hit2) << (GRP & 63u) : 0ull); \
        if (_m == 0) {                                                                        \
            /* no code matches: only possible for the lone 1-bit code (incomplete set) */     \
            INF_NEED((C)->max_len);                                                           \
            uint32_t _d;                                                                      \
            INF_TAKE(_d, (C)->max_len);                                                       \
            (void)_d;                                                                         \
            (OUTSYM) = -2;                                                                       \
            break;                                                                            \
        }                                                                                     \
        const uint32_t _len = (uint32_t)CTZ64(_m);                                            \
        if (br.bits < _len)                                                                   \
            INF_ON_EXHAUST;                                                                   \
        const uint32_t _code = _r >> (15 - _len);                                             \
        (OUTSYM) = (int)(C)->sym[(C)->offs[_len] + (_code - (C)->first[_len])];                  \
        br.hold >>= _len;                                                                     \
        br.bits -= _len;                                                                      \
    } while (0)

/* The same for the two codes of the symbol loop, with each length's first code, count and
 * symbol offset held by lane `length` in registers (FC = first | count << 16, OF = offs):
 * one LDS read per symbol instead of five. */
#define INF_DECODE_R(C, FC, OF, FC2, OF2, MAXLEN, WHICH, OUTSYM)                               \
    do {                                                                                      \
        /* topped up to more than 32 bits here (or to the end of the input), the code and the \
         * extra bits behind it (15 + 13 at most) need no second look at the input */         \
        if (br.bits <= 32)                                                                    \
            INF_REFILL();                                                                     \
        const uint32_t _peek = (uint32_t)br.hold & 0x7fffu;                                   \
        const uint32_t _r = BREV32(_peek) >> 17; /* the 15 bits MSB first */                  \
        LANEVAR(int, _hit);                                                                   \
        LANEVAR(int, _hit2);                                                                  \
        FOR_GLANES                                                                             \
        {                                                                                     \
            const uint32_t _l = (uint32_t)GLANE;                                               \
            const uint32_t _c = _r >> ((15u - _l) & 31u);                                     \
            LV(_hit) = _l >= 1u && _l <= (MAXLEN) &&                                          \
                       (uint32_t)(_c - (LV(FC) & 0xffffu)) < (LV(FC) >> 16);                  \
            const uint32_t _l2 = _l + GRP;                                                    \
            const uint32_t _c2 = _r >> ((15u - _l2) & 31u);                                   \
            LV(_hit2) = INF_SLOTS > 1 && _l2 <= (MAXLEN) &&                                   \
                        (uint32_t)(_c2 - (LV(FC2) & 0xffffu)) < (LV(FC2) >> 16);              \
        }                                                                                     \
        const uint64_t _m = GBALLOT(_hit) | (INF_SLOTS > 1 ? GBALLOT(_hit2) << (GRP & 63u) : 0ull); \
        if (_m == 0) /* not a code of the set, or the set is empty (MAXLEN 0): dealt with outside */ \
            INF_LEAVE(WHICH);                                                                 \
        const uint32_t _len = (uint32_t)CTZ64(_m);                                            \
        if (br.bits < _len)                                                                   \
            INF_ON_EXHAUST;                                                                   \
        const uint32_t _code = _r >> (15 - _len);                                             \
        const uint32_t _ol = (INF_SLOTS > 1 && _len >= GRP) ? GREADLANE(OF2, _len - GRP) : GREADLANE(OF, _len); \
        const uint32_t _fl = (INF_SLOTS > 1 && _len >= GRP) ? GREADLANE(FC2, _len - GRP) : GREADLANE(FC, _len); \
        (OUTSYM) = (int)GUNI((C)->sym[_ol + (_code - (_fl & 0xffffu))]);                      \
        br.hold >>= _len;                                                                     \
        br.bits -= _len;                                                                      \
    } while (0)

    /* HEAD .. HCRC, reference src/inflate.c:740-954 */
    if (wb && (wb < 8 || wb > 15)) {
        rc = -2;
        goto done;
    }
    if (resumed) {
        /* zsc_uncompress answers Z_DATA_ERROR with inflateSync (src/zsc_uncompr.c:109-125,
         * src/inflate.c:1547-1604): find the next 00 00 FF FF -- first in what was left of the
         * bit buffer, then in the input -- and decode on from there as a raw stream: mode =
         * TYPE, empty window, the totals and the check value carry on */
        pos = out_base = GUNI(rs->out_pos);
        data_errors = GUNI(rs->errors);
        gzip = (int)GUNI(rs->gzip);
        const uint64_t sy0 = ((uint64_t)GUNI(rs->sy_hi) << 32) | GUNI(rs->sy_lo);
        const uint32_t rb0 = GUNI(rs->sy_rb);
        const uint32_t nin = (uint32_t)((sy0 + rb0) >> 3); /* the reference's next_in */
        if (nin >= n && rb0 < 8u) {                         /* :1562-1565 */
            rc = INF_BUF;
            INF_SEEK(n < nin ? n : nin);
            goto done;
        }
        const uint32_t taken = inf_sync_search(src, n, sy0, rb0, nin);
        if (taken == 0xffffffffu) {
            rc = INF_DATA; /* the search used up all the input (:1585-1593) */
            INF_SEEK(n);
            goto done;
        }
        INF_SEEK(nin + taken);
    } else if (wrap) {
        INF_NEED(16);
        const uint32_t hw = (uint32_t)br.hold & 0xffffu;
        if ((wrap & 2) && hw == 0x8b1fu) {
            uint32_t t, flags;
            gzip = 1;
            INF_TAKE(t, 16);
            INF_NEED(16);
            INF_TAKE(flags, 16);
            if ((flags & 0xff) != 8 || (flags & 0xe000))
                INF_BADX(16, 16); /* the flags word is still in hold, :771-779 */
            INF_NEED(32);
            INF_TAKE(t, 32);
            INF_NEED(16);
            INF_TAKE(t, 16);
            if (flags & 0x0400) {
                uint32_t xlen;
                INF_NEED(16);
                INF_TAKE(xlen, 16);
                for (uint32_t k = 0; k < xlen; k++) {
                    INF_NEED(8);
                    INF_TAKE(t, 8);
                }
            }
            if (flags & 0x0800) {
                do {
                    INF_NEED(8);
                    INF_TAKE(t, 8);
                } while (t != 0);
            }
            if (flags & 0x1000) {
                do {
                    INF_NEED(8);
                    INF_TAKE(t, 8);
                } while (t != 0);
            }
            if (flags & 0x0200) {
                const uint32_t upto = (uint32_t)(BR_USED >> 3);
                uint32_t got;
                INF_NEED(16);
                INF_TAKE(got, 16);
                if ((wrap & 4) && got != (INF_CK(crc32_tx)<1>(src, upto, lds->cktab, INF_CKX(lds)) & 0xffffu))
                    INF_BADX(16, 16); /* :944-950 */
            }
        } else {
            uint32_t t = 0;
            (void)t;
            if (!(wrap & 1) || ((((hw & 0xff) << 8) + (hw >> 8)) % 31u))
                INF_BADX(0, 16); /* nothing dropped yet, :746-756 */
            if ((hw & 0xf) != 8)
                INF_BADX(0, 16);
            const uint32_t len = ((hw >> 4) & 0xf) + 8;
            const uint32_t wbits_eff = wb ? (uint32_t)wb : len;
            if (len > 15 || len > wbits_eff)
                INF_BADX(-4, 12); /* after DROPBITS(4), :757-764 */
            dmax = 1u << len;
            INF_TAKE(t, 16);
            if (hw & 0x2000) {
                /* preset dictionary: the reference returns Z_NEED_DICT from the DICT state
                 * without its exit bookkeeping, so nothing counts as consumed */
                INF_NEED(32);
                rc = INF_NEED_DICT;
                goto done;
            }
        }
    }

    /* blocks */
    for (;;) {
        uint32_t last, type;
        INF_NEED(3);
        INF_TAKE(last, 1);
        INF_TAKE(type, 2);
        if (type == 3)
            INF_BAD;
        if (type == 0) {
            uint32_t t, v;
            INF_TAKE(t, br.bits & 7u);
            (void)t;
            INF_NEED(32);
            INF_TAKE(v, 32);
            if ((v & 0xffff) != ((v >> 16) ^ 0xffff))
                INF_BADX(32, 32); /* LEN/NLEN still in hold, :1011-1016 */
            uint32_t len = v & 0xffff;
            /* the remaining bytes of the bit buffer belong to the stored data */
            const uint32_t at = (uint32_t)(BR_USED >> 3); /* input offset of the first data byte */
            uint32_t can = len;
            int short_in = 0, short_out = 0;
            if (can > n - at) {
                can = n - at;
                short_in = 1;
            }
            if (can > cap - pos) {
                can = cap - pos;
                short_out = 1;
            }
            /* 256 bytes per step, into the output and the ring */
            for (uint32_t k = 0; k < can; k += 256u) {
                const uint32_t step = can - k < 256u ? can - k : 256u;
                FOR_GLANES
                {
                    for (uint32_t j = (uint32_t)GLANE; j < step; j += GRP)
                    {
                        const uint8_t b = src[at + k + j];
                        lds->stage[(pos + j) & (INF_STAGE - 1)] = b;
                        dst[pos + j] = b;
                    }
                }
                WAVE_SYNC();
                pos += step;
            }
            INF_SEEK(at + can);
            if (short_in || short_out) {
                rc = INF_BUF; /* COPY state leaves with nothing more to do, src/inflate.c:1037-1039 */
                goto done;
            }
        } else {
            if (type == 1) {
                FOR_GLANES
                {
                    for (int s = GLANE; s < 288; s += GRP)
                        lds->lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                }
                WAVE_SYNC();
                (void)inf_build(&lds->lit, lds->lens, 288, 1);
                FOR_GLANES
                {
                    for (int s = GLANE; s < 32; s += GRP)
                        lds->lens[s] = 5;
                }
                WAVE_SYNC();
                (void)inf_build(&lds->dist, lds->lens, 32, 2);
            } else {
                uint32_t nlen, ndist, ncode;
                INF_NEED(14);
                INF_TAKE(nlen, 5);
                INF_TAKE(ndist, 5);
                INF_TAKE(ncode, 4);
                nlen += 257;
                ndist += 1;
                ncode += 4;
                if (nlen > 286 || ndist > 30)
                    INF_BAD;
                FOR_GLANES
                {
                    for (int s = GLANE; s < 19; s += GRP)
                        lds->lens[s] = 0;
                }
                WAVE_SYNC();
                for (uint32_t i = 0; i < ncode; i++) {
                    uint32_t v;
                    INF_NEED(3);
                    INF_TAKE(v, 3);
                    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    ON_GLANE0 { lds->lens[order[i]] = (uint8_t)v; }
                    WAVE_SYNC();
                }
                if (inf_build(&lds->cl, lds->lens, 19, 0))
                    INF_BAD;
                uint32_t have = 0;
                while (have < nlen + ndist) {
                    int sym;
                    INF_DECODE(&lds->cl, sym);
                    if (sym < 0)
                        sym = 0; /* all-zero code-length code: CODELENS reads the marker's val 0, bits 1 (:1105-1114) */
                    if (sym < 16) {
                        ON_GLANE0 { lds->lens[have] = (uint8_t)sym; }
                        WAVE_SYNC();
                        have++;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        INF_NEED(2);
                        if (have == 0) {
                            /* NEEDBITS(here.bits + 2) may have pulled one byte more (:1116-1123) */
                            const uint32_t padb = (8u - (uint32_t)(BR_USED & 7u)) & 7u;
                            INF_BADX(0, padb >= 2u ? padb : padb + 8u);
                        }
                        val = lds->lens[have - 1];
                        INF_TAKE(rep, 2);
                        rep += 3;
                    } else if (sym == 17) {
                        INF_NEED(3);
                        INF_TAKE(rep, 3);
                        rep += 3;
                    } else {
                        INF_NEED(7);
                        INF_TAKE(rep, 7);
                        rep += 11;
                    }
                    if (have + rep > nlen + ndist)
                        INF_BAD;
                    FOR_GLANES
                    {
                        for (uint32_t k = (uint32_t)GLANE; k < rep; k += GRP)
                            lds->lens[have + k] = (uint8_t)val;
                    }
                    WAVE_SYNC();
                    have += rep;
                }
                if (lds->lens[256] == 0)
                    INF_BAD;
                if (inf_build(&lds->lit, lds->lens, (int)nlen, 1))
                    INF_BAD;
                if (inf_build(&lds->dist, lds->lens + nlen, (int)ndist, 2))
                    INF_BAD;
            }
            /* symbols */
            WAVE_SYNC();
            LANEVAR(uint32_t, lfc);
            LANEVAR(uint32_t, lof);
            LANEVAR(uint32_t, dfc);
            LANEVAR(uint32_t, dof);
            LANEVAR(uint32_t, lfc2); /* (groups of fewer than 16 lanes: the lengths GLANE + GRP) */
            LANEVAR(uint32_t, lof2);
            LANEVAR(uint32_t, dfc2);
            LANEVAR(uint32_t, dof2);
            FOR_GLANES
            {
                const int l = GLANE & 15, l2 = (GLANE + (int)GRP) & 15;
                LV(lfc) = (uint32_t)lds->lit.first[l] | ((uint32_t)lds->lit.count[l] << 16);
                LV(lof) = lds->lit.offs[l];
                LV(dfc) = (uint32_t)lds->dist.first[l] | ((uint32_t)lds->dist.count[l] << 16);
                LV(dof) = lds->dist.offs[l];
                LV(lfc2) = (uint32_t)lds->lit.first[l2] | ((uint32_t)lds->lit.count[l2] << 16);
                LV(lof2) = lds->lit.offs[l2];
                LV(dfc2) = (uint32_t)lds->dist.first[l2] | ((uint32_t)lds->dist.count[l2] << 16);
                LV(dof2) = lds->dist.offs[l2];
            }
            const uint32_t lmax = GUNI(lds->lit.max_len), lempty = GUNI(lds->lit.empty);
            const uint32_t dmaxlen = GUNI(lds->dist.max_len), dempty = GUNI(lds->dist.empty);
            /* The symbol loop.  Its shape matters more than its instruction count suggests: with the
             * error exits as gotos to the function's exit code the compiler threaded a dispatch
             * variable and copies of the live state through every join (~40 % of the vector
             * instructions); now every way out of the loop is the ONE exit below with an event
             * code, output bytes go straight to dst, and what is rare (a code that is none, the
             * exit bookkeeping) happens outside.  Literal runs are an inner loop, so that the
             * groups of a wave that have reached a match go through the match path together.
             * (Measured and dropped: an inflate_fast-style second copy of the loop without the
             * exhaustion tests -- more registers and code than the tests cost.) */
            uint32_t ev = 0; /* why the symbol loop was left: 0 = end of block, else class << 16 | line */
#pragma push_macro("INF_ON_EXHAUST")
#pragma push_macro("INF_ON_FULL")
#pragma push_macro("INF_BAD")
#undef INF_ON_EXHAUST
#undef INF_ON_FULL
#undef INF_BAD
#define INF_LEAVE(cls)                              \
    do {                                            \
        ev = ((uint32_t)(cls) << 16) | __LINE__;    \
        goto sym_exit;                              \
    } while (0)
#define INF_ON_EXHAUST INF_LEAVE(1)
#define INF_ON_FULL INF_LEAVE(2)
#define INF_BAD INF_LEAVE(3)
            for (;;) {
                int sym;
                for (;;) {
                    INF_DECODE_R(&lds->lit, lfc, lof, lfc2, lof2, lmax, 4, sym);
                    if (sym >= 256) /* a length or the end of the block */
                        break;
                    if (pos >= cap)
                        INF_ON_FULL;
                    ON_GLANE0
                    {
                        lds->stage[pos & (INF_STAGE - 1)] = (uint8_t)sym;
                        dst[pos] = (uint8_t)sym;
                    }
                    WAVE_SYNC();
                    pos++;
                }
                if (sym == 256)
                    break;
                /* What can end the symbol before its copy is tested once per code, in straight-line
                 * code; which of the reasons it was, in the reference's order, is sorted out in the
                 * (cold) block behind the test. */
                uint32_t c = (uint32_t)sym - 257u, xb, ex, len;
                xb = (c < 8 || c >= 28) ? 0u : (c - 4u) >> 2;
                if ((sym > 285) | (br.bits < xb)) {
                    if (sym > 285)
                        INF_BAD;
                    INF_ON_EXHAUST; /* (the top-up before the code took what input there was) */
                }
                INF_TAKE(ex, xb);
                len = c < 8 ? c + 3u : c == 28 ? 258u : ((4u + ((c - 4u) & 3u)) << ((c - 4u) >> 2)) + 3u + ex;
                int ds;
                INF_DECODE_R(&lds->dist, dfc, dof, dfc2, dof2, dmaxlen, 5, ds);
                const uint32_t dsu = (uint32_t)ds; /* 0..31 */
                xb = dsu < 4 ? 0u : (dsu >> 1) - 1u;
                const uint32_t dist = (dsu < 4 ? dsu : (2u + (dsu & 1u)) << ((dsu >> 1) - 1u)) + 1u +
                                      ((uint32_t)br.hold & ((1u << xb) - 1u));
                if ((dsu > 29u) | (br.bits < xb) | (dist > dmax) | (pos >= cap) | (dist > pos - out_base)) {
                    if (dsu > 29u)
                        INF_BAD;
                    if (br.bits < xb)
                        INF_ON_EXHAUST;
                    INF_TAKE(ex, xb);
                    if (dist > dmax) /* DISTEXT, :1266-1272 */
                        INF_BAD;
                    if (pos >= cap) /* MATCH leaves on a full output before it looks at the distance (:1277) */
                        INF_ON_FULL;
                    INF_BAD; /* :1279-1288; nothing behind an inflateSync can be copied */
                }
                INF_TAKE(ex, xb);
                uint32_t can = len;
                if (can > cap - pos)
                    can = cap - pos;
                /* lane-parallel copy: byte i comes from pos - dist + (i mod dist).  i <= 257, so the
                 * quotient is floor((i + 0.5) / dist) in single precision -- (i + 0.5) / dist is never
                 * within 1 / (2 dist) >= 2e-3 of an integer when dist <= i, and below 1 otherwise,
                 * far more than the error of the reciprocal -- a handful of instructions where the
                 * integer division is ~30 */
                const float rdist = RCP_F32((float)dist);
                for (uint32_t k = 0; k < can; k += GRP) {
                    FOR_GLANES
                    {
                        uint32_t i = k + (uint32_t)GLANE;
                        if (i < can) {
                            const uint32_t q = (uint32_t)(((float)i + 0.5f) * rdist);
                            uint32_t s = pos - dist + (i - q * dist);
                            /* every byte less than INF_STAGE before the end of this copy is still in
                             * the ring; older ones are read back from the output in global memory */
                            uint8_t b = s + INF_STAGE >= pos + can ? lds->stage[s & (INF_STAGE - 1)] : dst[s];
                            lds->stage[(pos + i) & (INF_STAGE - 1)] = b;
                            dst[pos + i] = b;
                        }
                    }
                    WAVE_SYNC();
                }
                pos += can;
                if (can < len)
                    INF_ON_FULL;
            }
        sym_exit:
#pragma pop_macro("INF_BAD")
#pragma pop_macro("INF_ON_FULL")
#pragma pop_macro("INF_ON_EXHAUST")
#undef INF_LEAVE
            if (ev != 0) {
                const uint32_t cls = ev >> 16;
                if (cls == 1u)
                    INF_ON_EXHAUST;
                if (cls == 2u)
                    INF_ON_FULL;
                if (cls >= 4u) {
                    /* the bits were no code of the literal/length (4) or distance (5) set: the
                     * reference's table has invalid-code markers there, of length 1 for an empty
                     * set (src/inftrees.c:150-158) and of the longest length otherwise, and drops
                     * them before it looks at the entry (:1217-1236) */
                    const uint32_t drop = cls == 4u ? (lempty ? 1u : lmax) : (dempty ? 1u : dmaxlen);
                    uint32_t dropped;
                    INF_NEED(drop);
                    INF_TAKE(dropped, drop);
                    (void)dropped;
                }
                /* a data error (INF_BAD where it happened; the bit reader has not moved since) */
                sy_start = BR_USED;
                sy_rb = (8u - (uint32_t)(BR_USED & 7u)) & 7u;
                fail_line = ev & 0xffffu;
                goto bad;
            }
        }
        if (last)
            break;
    }

    /* CHECK / LENGTH, reference src/inflate.c:1322-1354 */
    {
        uint32_t t;
        INF_TAKE(t, br.bits & 7u);
        (void)t;
        if (wrap) {
            uint32_t v;
            INF_NEED(32);
            INF_TAKE(v, 32);
            if (wrap & 4) {
#ifndef ZSC_WAVE_EMU
                __threadfence_block();
#endif
                const uint32_t want = gzip ? INF_CK(crc32_tx)<1>(dst, pos, lds->cktab, INF_CKX(lds)) : INF_CK(adler32)(dst, pos);
                const uint32_t got = gzip ? v : ((v >> 24) | ((v >> 8) & 0xff00u) | ((v & 0xff00u) << 8) | (v << 24));
                if (got != want)
                    INF_BADX(32, 32); /* :1333-1339 */
            }
            if (gzip) {
                INF_NEED(32);
                INF_TAKE(v, 32);
                if (v != pos - out_base) /* state->total restarts at an inflateSync */
                    INF_BADX(32, 32); /* :1347-1351 */
            }
        }
        rc = data_errors ? INF_DATA : INF_END; /* src/zsc_uncompr.c:149-152 */
        goto done;
    }

bad:
    /* a data error: hand the state inflateSync starts from to the next entry */
    data_errors++;
    ON_GLANE0
    {
        rs->state = 1;
        rs->out_pos = pos;
        rs->errors = data_errors;
        rs->gzip = (uint32_t)gzip;
        rs->sy_lo = (uint32_t)sy_start;
        rs->sy_hi = (uint32_t)(sy_start >> 32);
        rs->sy_rb = sy_rb;
    }
    WAVE_SYNC();
    return 1;

done:
    ON_GLANE0
    {
        uint32_t used_bytes = (uint32_t)((BR_USED + 7u) >> 3);
        if (exhausted || used_bytes > n)
            used_bytes = n;
        res->status = rc == INF_END ? 0 : rc;
        res->out_len = rc == INF_NEED_DICT ? 0u : pos;
        res->consumed = rc == INF_NEED_DICT ? 0u : used_bytes;
        res->pad = fail_line;
        rs->state = 2;
    }
    return 0;
#undef INF_REFILL
#undef INF_NEED
#undef INF_TAKE
#undef INF_BAD
#undef INF_BADX
#undef INF_SEEK
#undef INF_DECODE
#undef INF_DECODE_R
}


/* zsc_uncompress for one stream on the host emulation: inflate again after every recovered
 * data error (the GPU runtime relaunches the kernel instead, see inflate_stream) */
DEV void inflate_with_resync(const InfJob &job, InfLds *lds, InfResult *res)
{
    InfResume rs;
    rs.state = 0;
    rs.out_pos = rs.errors = rs.gzip = rs.sy_lo = rs.sy_hi = rs.sy_rb = 0;
    for (uint32_t round = 0; round < job.n / 4u + 2u; round++) {
        if (!inflate_stream(job, lds, res, &rs))
            return;
    }
}

/* back to whole-wave groups for whatever is compiled after this */
#undef ZSC_GROUP
#define ZSC_GROUP 64
#include "wave_group.h"

#endif
