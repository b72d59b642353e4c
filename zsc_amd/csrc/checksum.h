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
 * inverted): every lane runs the table-driven byte loop over its own contiguous
 * 1/64th of the buffer, then lane 0 folds the 64 partial CRCs left to right with
 * crc(A||B) = crc(A) * x^(8|B|) mod P  xor  crc(B)  (GF(2) multiply).
 */
#ifndef ZSC_CHECKSUM_H
#define ZSC_CHECKSUM_H

#include "wave.h"

#define CK_BASE 65521u

typedef struct {
    uint32_t table[256];
    uint32_t part[WAVE];
    uint32_t plen[WAVE];
} CkLds;

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

/* a * b mod P over GF(2), reflected representation (x^0 is bit 31) */
DEV uint32_t ck_mulmod(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0)
                break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

/* x^(8*len) mod P */
DEV uint32_t ck_xpow8(uint32_t len)
{
    uint32_t r = 1u << 31;   /* x^0 */
    uint32_t sq = 1u << 23;  /* x^8 */
    while (len) {
        if (len & 1u)
            r = ck_mulmod(sq, r);
        sq = ck_mulmod(sq, sq);
        len >>= 1;
    }
    return r;
}

DEV uint32_t ck_crc32(const uint8_t *in, uint32_t n, CkLds *lds)
{
    for (int i = 0; i < 256; i += WAVE) {
        FOR_LANES
        {
            uint32_t c = (uint32_t)(i + LANE);
            for (int k = 0; k < 8; k++)
                c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            lds->table[i + LANE] = c;
        }
    }
    WAVE_SYNC();
    const uint32_t seg = (n + WAVE - 1) / WAVE;
    FOR_LANES
    {
        uint32_t lo = (uint32_t)LANE * seg;
        uint32_t hi = lo + seg < n ? lo + seg : n;
        uint32_t c = 0xffffffffu;
        for (uint32_t i = lo; i < hi; i++)
            c = lds->table[(c ^ in[i]) & 0xffu] ^ (c >> 8);
        lds->part[LANE] = ~c;
        lds->plen[LANE] = hi > lo ? hi - lo : 0u;
    }
    WAVE_SYNC();
    uint32_t acc = 0;
    ON_LANE0
    {
        for (int l = 0; l < WAVE; l++) {
            uint32_t len = lds->plen[l];
            if (len == 0)
                continue;
            acc = ck_mulmod(ck_xpow8(len), acc) ^ lds->part[l];
        }
        lds->part[0] = acc;
    }
    WAVE_SYNC();
    return lds->part[0];
}

#endif
