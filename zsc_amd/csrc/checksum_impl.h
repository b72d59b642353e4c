/*
 * checksum_impl.h -- the bodies of the checksum routines, written for a group of lanes
 * (wave_group.h).  checksum.h includes this once for whole-wave groups (ck_adler32,
 * ck_crc32_t); a kernel source that runs several units per wave includes it again under its
 * own group width with CK_FN giving the routines another name.  No include guard.
 */
DEV uint32_t CK_FN(adler32)(const uint8_t *in, uint32_t n)
{
    LANEVAR(uint64_t, sd); /* sum of bytes */
    LANEVAR(uint64_t, si); /* sum of (index mod BASE) * byte */
    LANEVAR(uint32_t, im); /* index of the lane's current piece, mod BASE */
    FOR_GLANES
    {
        LV(sd) = 0;
        LV(si) = 0;
        LV(im) = ((uint32_t)GLANE * 16u) % CK_BASE;
    }
    for (uint32_t base = 0; base < n; base += GRP * 16) {
        FOR_GLANES
        {
            uint32_t a = base + (uint32_t)GLANE * 16u;
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
            uint32_t nx = LV(im) + (GRP * 16u) % CK_BASE;
            LV(im) = nx >= CK_BASE ? nx - CK_BASE : nx;
        }
    }
    const uint64_t SD = GSUM64(sd) % CK_BASE;
    const uint64_t SI = GSUM64(si) % CK_BASE;
    const uint64_t nm = n % CK_BASE;
    const uint32_t A = (uint32_t)((1u + SD) % CK_BASE);
    const uint32_t B = (uint32_t)((nm + nm * SD + CK_BASE - SI) % CK_BASE);
    return (B << 16) | A;
}

template <int NT>
DEV uint32_t CK_FN(crc32_tx)(const uint8_t *in, uint32_t n, uint32_t (*table)[256], uint32_t *xch)
{
    for (int i = 0; i < 256; i += GRP) {
        FOR_GLANES
        {
            uint32_t c = (uint32_t)(i + GLANE);
            for (int k = 0; k < 8; k++)
                c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[0][i + GLANE] = c;
        }
    }
    WAVE_SYNC();
    for (int t = 1; t < NT; t++) {
        for (int i = 0; i < 256; i += GRP) {
            FOR_GLANES
            {
                const uint32_t c = table[t - 1][i + GLANE];
                table[t][i + GLANE] = table[0][c & 0xffu] ^ (c >> 8);
            }
        }
        WAVE_SYNC();
    }
    /* segment length: the smallest power of two >= 16 with 64 segments covering the buffer */
    uint32_t k = 4;
    while (k < 31u && ((uint64_t)GRP << k) < n)
        k++;
    const uint32_t S = 1u << k;
    const uint32_t last = n ? (n - 1u) >> k : 0u;     /* the lane of the last byte */
    const uint32_t len_last = n - (last << k);        /* 1..S (0 for the empty buffer) */
    LANEVAR(uint32_t, c);
    FOR_GLANES
    {
        const uint64_t lo64 = (uint64_t)GLANE << k;
        uint32_t r = GLANE == 0 ? 0xffffffffu : 0u;
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
                        r = table[3 % NT][r & 0xffu] ^ table[2 % NT][(r >> 8) & 0xffu] ^
                            table[1 % NT][(r >> 16) & 0xffu] ^ table[0][r >> 24];
                    } else {
                        for (int b = 0; b < 4; b++)
                            r = table[0][r & 0xffu] ^ (r >> 8);
                    }
                }
            }
            for (; i < hi; i++)
                r = table[0][(r ^ in[i]) & 0xffu] ^ (r >> 8);
        }
        LV(c) = r;
    }
    /* the lanes before `last` hold full segments: bring them to the top of the wave, the last
     * full one to lane 63, zeros below */
    const uint32_t shift = (GRP - 1u) - (last ? last - 1u : 0u); /* lanes to move up (last >= 1) */
    FOR_GLANES { xch[GLANE] = LV(c); }
    WAVE_SYNC();
    const uint32_t c_last = GUNI(xch[last]);
    LANEVAR(uint32_t, v);
    FOR_GLANES
    {
        const uint32_t l = (uint32_t)GLANE;
        LV(v) = (last != 0u && l >= shift) ? xch[l - shift] : 0u;
    }
    WAVE_SYNC();
    /* log-steps: the value at the right end of a block of 2d lanes becomes left * X^d + right */
    for (uint32_t m = 0; (1u << m) < GRP; m++) {
        const uint32_t d = 1u << m;
        const uint32_t xd = ck_x2n(3u + k + m); /* X^d = x^(8 S d) */
        FOR_GLANES { xch[GLANE] = LV(v); }
        WAVE_SYNC();
        FOR_GLANES
        {
            const uint32_t l = (uint32_t)GLANE;
            const uint32_t left = l >= d ? xch[l - d] : 0u;
            const uint32_t comb = ck_mulmod(left, xd) ^ LV(v);
            if ((l & (2u * d - 1u)) == 2u * d - 1u)
                LV(v) = comb;
        }
        WAVE_SYNC();
    }
    /* x^(8 len_last): the product of x^(2^j) over the set bits j of 8 len_last */
    LANEVAR(uint32_t, f);
    FOR_GLANES
    {
        const uint64_t bits = (uint64_t)len_last * 8u;
        uint32_t prod = 0x80000000u; /* x^0 */
        for (uint32_t j = (uint32_t)GLANE; j < 40u; j += GRP)
            prod = ((bits >> j) & 1ull) ? ck_mulmod(prod, ck_x2n(j)) : prod;
        LV(f) = prod;
    }
    for (uint32_t d = 1; d < GRP; d <<= 1) {
        FOR_GLANES { xch[GLANE] = LV(f); }
        WAVE_SYNC();
        FOR_GLANES { LV(f) = ck_mulmod(LV(f), xch[(uint32_t)GLANE ^ d]); }
        WAVE_SYNC();
    }
    /* (every lane ends up with the whole product) */
    FOR_GLANES { xch[GLANE] = GLANE == 0 ? LV(f) : LV(v); }
    WAVE_SYNC();
    const uint32_t pw = GUNI(xch[0]), vtop = GUNI(xch[GRP - 1u]);
    WAVE_SYNC();
    const uint32_t total = ck_mulmod(vtop, pw) ^ c_last;
    return ~total;
}


/* (the tables may be shared by the groups of a wave: every group writes all of them, with the
 * same values, before it reads any; the exchange area is the group's own) */
template <int NT>
DEV uint32_t CK_FN(crc32_t)(const uint8_t *in, uint32_t n, CkLdsT<NT> *lds)
{
    return CK_FN(crc32_tx)<NT>(in, n, lds->table, lds->x);
}
