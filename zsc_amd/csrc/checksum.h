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

#define CK_FN(name) ck_##name
#include "checksum_impl.h"
#undef CK_FN

DEV uint32_t ck_crc32(const uint8_t *in, uint32_t n, CkLds *lds)
{
    return ck_crc32_t<4>(in, n, lds);
}

#endif
