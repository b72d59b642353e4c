/*
 * checksum.h -- kernel 0: Adler-32 / CRC-32 of every input buffer, one wavefront
 * per buffer.
 *
 * Adler-32 (reference src/adler32.c:56-131) is a = 1 + sum(d_i), b = sum of the
 * running a, both mod 65521.  Unrolled over the whole buffer,
 *     b = n + sum((n - i) * d_i),
 * so 64 lanes can each take 16-byte pieces, keep sum(d) and sum(i*d) (the index
 * reduced mod 65521 so 64-bit sums cannot overflow), and one wave reduction at
 * the end finishes the job -- no serial dependence, one coalesced pass.
 *
 * CRC-32 (reference src/crc32.c:502-593, polynomial 0xEDB88320, pre/post
 * inverted): every lane runs the slicing-by-4 loop over its own power-of-two sized
 * segment of the buffer, 16 bytes per load; the 64 remainders are combined with
 * r(A||B) = r(A) * x^(8|B|) mod P  xor  r(B) in log-steps over the lanes, every lane
 * doing its GF(2) multiplication at the same time (ck_crc32_t).
 */
#ifndef ZSC_CHECKSUM_H
#define ZSC_CHECKSUM_H

#include "wave.h"

#define CK_BASE 65521u

DEV uint32_t ck_adler32(const uint8_t *in, uint32_t n)
{
    LANEVAR(uint64_t, sd); /* sum of bytes */
    LANEVAR(uint64_t, si); /* sum of (index mod BASE) * byte */
    LANEVAR(uint32_t, im); /* index of the lane's current piece, mod BASE */
    FOR_LANES
    {
        LV(sd) = 0;
        LV(si) = 0;
        LV(im) = ((uint32_t)LANE * 16u) % CK_BASE;
    }
    for (uint32_t base = 0; base < n; base += WAVE * 16) {
        FOR_LANES
        {
            uint32_t a = base + (uint32_t)LANE * 16u;
            if (a < n) {
                uint8_t b[16];
                if (a + 16 <= n) {
                    COPY16(b, in + a);
                } else {
                    for (uint32_t j = 0; j < 16; j++)
                        b[j] = a + j < n ? in[a + j] : (uint8_t)0;
                }
                uint32_t s1 = 0, s2 = 0;
                for (uint32_t j = 0; j < 16; j++) {
                    s1 += b[j];
                    s2 += j * b[j];
                }
                LV(sd) += s1;
                LV(si) += (uint64_t)LV(im) * s1 + s2;
            }
            uint32_t nx = LV(im) + (WAVE * 16u) % CK_BASE;
            LV(im) = nx >= CK_BASE ? nx - CK_BASE : nx;
        }
    }
    const uint64_t SD = WAVE_SUM(sd) % CK_BASE;
    const uint64_t SI = WAVE_SUM(si) % CK_BASE;
    const uint64_t nm = n % CK_BASE;
    const uint32_t A = (uint32_t)((1u + SD) % CK_BASE);
    const uint32_t B = (uint32_t)((nm + nm * SD + CK_BASE - SI) % CK_BASE);
    return (B << 16) | A;
}

/* a * b mod P over GF(2), reflected representation (x^0 is bit 31); a fixed 32 steps, no
 * branches, so that 64 lanes can each do their own multiplication in step */
DEV uint32_t ck_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
}

/* x^(2^j) mod P, j = 0..31 (the sequence repeats from there: x^(2^32) = x) */
DEV uint32_t ck_x2n(uint32_t j)
{
    static const uint32_t t[32] = {
        0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u,
        0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu,
        0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu,
        0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};
    return t[j & 31u];
}

/* CRC-32 of a buffer by one wavefront (reference crc32_z, src/crc32.c:502-593: polynomial
 * 0xEDB88320, register preset to and finally inverted with 0xffffffff).
 *
 * The buffer is cut into 64 segments of S bytes, S a power of two, one per lane.  Every lane
 * runs the table-driven loop over its own segment, 16 bytes per load (NT = 4: slicing-by-4,
 * four 1 KiB tables in LDS, four look-ups per four bytes, :563-593; NT = 1: the byte loop,
 * :517-525, for callers that are short of LDS).  The 64 remainders are then combined as
 * polynomials:  r(A || B) = r(A) * x^(8 |B|) + r(B)  mod P.  With all segments but the last
 * S bytes long the factors are powers of X = x^(8 S) = x^(2^k), a constant from a table:
 * log-steps over the lanes, each one multiplication per lane (ck_mulmod, all lanes in
 * step), then one multiplication by x^(8 len(last segment)) -- itself a product over the
 * set bits of the length, reduced over the lanes the same way.  The preset register is lane
 * 0's starting value. */
template <int NT>
struct CkLdsT {
    uint32_t table[NT][256];
    uint32_t x[WAVE]; /* lanes hand values to each other through here */
};
typedef CkLdsT<4> CkLds;

template <int NT>
DEV uint32_t ck_crc32_t(const uint8_t *in, uint32_t n, CkLdsT<NT> *lds)
{
    for (int i = 0; i < 256; i += WAVE) {
        FOR_LANES
        {
            uint32_t c = (uint32_t)(i + LANE);
            for (int k = 0; k < 8; k++)
                c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            lds->table[0][i + LANE] = c;
        }
    }
    WAVE_SYNC();
    for (int t = 1; t < NT; t++) {
        for (int i = 0; i < 256; i += WAVE) {
            FOR_LANES
            {
                const uint32_t c = lds->table[t - 1][i + LANE];
                lds->table[t][i + LANE] = lds->table[0][c & 0xffu] ^ (c >> 8);
            }
        }
        WAVE_SYNC();
    }
    /* segment length: the smallest power of two >= 16 with 64 segments covering the buffer */
    uint32_t k = 4;
    while (k < 31u && ((uint64_t)WAVE << k) < n)
        k++;
    const uint32_t S = 1u << k;
    const uint32_t last = n ? (n - 1u) >> k : 0u;     /* the lane of the last byte */
    const uint32_t len_last = n - (last << k);        /* 1..S (0 for the empty buffer) */
    LANEVAR(uint32_t, c);
    FOR_LANES
    {
        const uint64_t lo64 = (uint64_t)LANE << k;
        uint32_t r = LANE == 0 ? 0xffffffffu : 0u;
        if (lo64 < n) {
            const uint32_t lo = (uint32_t)lo64;
            const uint32_t hi = n - lo > S ? lo + S : n;
            uint32_t i = lo;
            for (; i + 16u <= hi; i += 16u) {
                uint32_t w[4];
                COPY16(w, in + i);
                for (int q = 0; q < 4; q++) {
                    r ^= w[q];
                    if (NT >= 4) {
                        r = lds->table[3 % NT][r & 0xffu] ^ lds->table[2 % NT][(r >> 8) & 0xffu] ^
                            lds->table[1 % NT][(r >> 16) & 0xffu] ^ lds->table[0][r >> 24];
                    } else {
                        for (int b = 0; b < 4; b++)
                            r = lds->table[0][r & 0xffu] ^ (r >> 8);
                    }
                }
            }
            for (; i < hi; i++)
                r = lds->table[0][(r ^ in[i]) & 0xffu] ^ (r >> 8);
        }
        LV(c) = r;
    }
    /* the lanes before `last` hold full segments: bring them to the top of the wave, the last
     * full one to lane 63, zeros below */
    const uint32_t shift = (WAVE - 1u) - (last ? last - 1u : 0u); /* lanes to move up (last >= 1) */
    FOR_LANES { lds->x[LANE] = LV(c); }
    WAVE_SYNC();
    const uint32_t c_last = UNI(lds->x[last]);
    LANEVAR(uint32_t, v);
    FOR_LANES
    {
        const uint32_t l = (uint32_t)LANE;
        LV(v) = (last != 0u && l >= shift) ? lds->x[l - shift] : 0u;
    }
    WAVE_SYNC();
    /* log-steps: the value at the right end of a block of 2d lanes becomes left * X^d + right */
    for (uint32_t m = 0; (1u << m) < WAVE; m++) {
        const uint32_t d = 1u << m;
        const uint32_t xd = ck_x2n(3u + k + m); /* X^d = x^(8 S d) */
        FOR_LANES { lds->x[LANE] = LV(v); }
        WAVE_SYNC();
        FOR_LANES
        {
            const uint32_t l = (uint32_t)LANE;
            const uint32_t left = l >= d ? lds->x[l - d] : 0u;
            const uint32_t comb = ck_mulmod(left, xd) ^ LV(v);
            if ((l & (2u * d - 1u)) == 2u * d - 1u)
                LV(v) = comb;
        }
        WAVE_SYNC();
    }
    /* x^(8 len_last): the product of x^(2^j) over the set bits j of 8 len_last */
    LANEVAR(uint32_t, f);
    FOR_LANES
    {
        const uint64_t bits = (uint64_t)len_last * 8u;
        uint32_t prod = 0x80000000u; /* x^0 */
        for (uint32_t j = (uint32_t)LANE; j < 40u; j += WAVE)
            prod = ((bits >> j) & 1ull) ? ck_mulmod(prod, ck_x2n(j)) : prod;
        LV(f) = prod;
    }
    for (uint32_t d = 1; d < WAVE; d <<= 1) {
        FOR_LANES { lds->x[LANE] = LV(f); }
        WAVE_SYNC();
        FOR_LANES { LV(f) = ck_mulmod(LV(f), lds->x[(uint32_t)LANE ^ d]); }
        WAVE_SYNC();
    }
    /* (every lane ends up with the whole product) */
    FOR_LANES { lds->x[LANE] = LANE == 0 ? LV(f) : LV(v); }
    WAVE_SYNC();
    const uint32_t pw = UNI(lds->x[0]), vtop = UNI(lds->x[WAVE - 1u]);
    WAVE_SYNC();
    const uint32_t total = ck_mulmod(vtop, pw) ^ c_last;
    return ~total;
}

DEV uint32_t ck_crc32(const uint8_t *in, uint32_t n, CkLds *lds)
{
    return ck_crc32_t<4>(in, n, lds);
}

#endif
