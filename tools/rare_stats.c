/* rare_stats.c -- development aid: what the "nearest improver through the rarest chain" search
 * would cost next to the reference's chain walk (src/deflate.c:1400-1518), on the positions the
 * lazy parse (src/deflate.c:1989-2122) really searches.  Window sliding approximated by
 * "distance <= MAX_DIST" (statistics only).
 *
 * A candidate that fails the pre-check is free in zsc, so longest_match(p, b0) is: the staircase of
 * nearest improvers (nearest c with LCP > best, then best = LCP(c), ...) cut off by nice_match,
 * provided the chain budget does not run out before the last step.  An improver at level `best`
 * shares best+1 bytes with p, so it lies on the chain of EVERY trigram p+j, j = 0..best-2: walk the
 * shortest one. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define MAXD 32506u
#define NIL 0xffffffffu
static uint8_t *in;
static uint32_t n, *head, *prev, *wcnt, *cntat;
static inline uint32_t H(uint32_t x) { return ((in[x] << 10) ^ (in[x + 1] << 5) ^ in[x + 2]) & 0x7fff; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int level = argc > 2 ? atoi(argv[2]) : 6;
    static const int cfgs[10][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32},{4,4,16,16},{8,16,32,32},{8,16,128,128},{8,32,128,256},{32,128,258,1024},{32,258,258,4096}};
    uint32_t good = cfgs[level][0], lazy = cfgs[level][1], nicec = cfgs[level][2], chain = cfgs[level][3];
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    in = malloc(n + 600); memset(in, 0, n + 600); if (fread(in, 1, n, f) != n) return 1;
    head = malloc(32768 * 4); prev = malloc(4 * (n + 1)); wcnt = calloc(32768, 4); cntat = calloc(n + 600, 4);
    for (int i = 0; i < 32768; i++) head[i] = NIL;
    /* chains are parse-independent at levels 4-9: build prev[] and the in-window bucket size at
     * every position up front */
    for (uint32_t x = 0; x + 2 < n; x++) {
        if (x > MAXD) { uint32_t y = x - MAXD - 1; wcnt[H(y)]--; }
        uint32_t h = H(x); prev[x] = head[h]; head[h] = x; cntat[x] = wcnt[h]; wcnt[h]++;
    }
    uint64_t calls = 0, ref_cand = 0, new_cand = 0, new_cand_top = 0, verif = 0, verif_calls = 0, bound_calls = 0, nsteps = 0, mism = 0;
    uint64_t new_fresh = 0, new_lazy = 0, ref_fresh = 0, ref_lazy = 0, headonly = 0;
    uint32_t p = 0, cur_len = 2, cur_at = 0; int pending = 0;
    while (p < n) {
        uint32_t look = n - p;
        uint32_t hh = look >= 3 ? prev[p] : NIL;
        uint32_t prev_len = cur_len, prev_at = cur_at; cur_len = 2;
        if (hh != NIL && hh != 0 && prev_len < lazy && p - hh <= MAXD) {
            calls++;
            uint32_t budget = chain, best = prev_len, nice = nicec, cap = look < 258 ? look : 258;
            if (prev_len >= good) budget >>= 2;
            const uint32_t B0 = budget;
            if (nice > look) nice = look;
            uint32_t c = hh, nc = 0;
            uint32_t limit = p > MAXD ? p - MAXD : 0;
            int bound = 0;
            uint32_t where = cur_at;
            if (best < look) for (;;) {
                nc++;
                const uint8_t *m = in + c, *s = in + p;
                if (m[best] == s[best] && m[best-1] == s[best-1] && m[0] == s[0] && m[1] == s[1]) {
                    uint32_t len = 2;
                    while (len < cap && m[len] == s[len]) len++;
                    if (len > best) { where = c; best = len; if (len >= nice) break; }
                    budget--;
                }
                c = prev[c];
                if (c == NIL || c <= limit) break;
                if (budget == 0) { bound = 1; break; }
            }
            ref_cand += nc; if (prev_len == 2) ref_fresh += nc; else ref_lazy += nc;
            bound_calls += bound;
            /* ---- the new scheme, unlimited budget ---- */
            {
                uint32_t b = prev_len, bnd = p, w2 = cur_at, cost = 0, cost_top = 0;
                const uint8_t *s = in + p;
                if (b < look) for (;;) {
                    /* rarest trigram among offsets 0..b-2 (b == 2: offset 0) */
                    uint32_t jbest = 0, lbest = cntat[p];
                    for (uint32_t j = 1; j + 2 <= b && p + j + 2 < n; j++)
                        if (cntat[p + j] < lbest) { lbest = cntat[p + j]; jbest = j; }
                    nsteps++;
                    /* walk chain(p + jbest) for the nearest c < bnd, c > limit, with LCP(c, p) > b */
                    uint32_t cc = prev[p + jbest], found = NIL, flen = 0, walked = 0, skipped = 0;
                    while (cc != NIL && cc >= jbest && cc - jbest > limit && !(cc - jbest == 0)) {
                        uint32_t q = cc - jbest;
                        if (q >= bnd) { skipped++; cc = prev[cc]; continue; }
                        walked++;
                        const uint8_t *m = in + q;
                        if (m[b] == s[b] && m[b - 1] == s[b - 1]) {
                            uint32_t len = 0;
                            while (len < cap && m[len] == s[len]) len++;
                            if (len > b) { found = q; flen = len; break; }
                        }
                        cc = prev[cc];
                    }
                    cost += walked; cost_top += walked + skipped;
                    if (found == NIL) break;
                    b = flen; w2 = found; bnd = found;
                    if (b >= nice) break;
                }
                new_cand += cost; new_cand_top += cost_top;
                if (prev_len == 2) new_fresh += cost; else new_lazy += cost;
                if (cost <= 1) headonly++;
                /* budget check: candidates of chain(p) nearer than the last record */
                if (b > prev_len) {
                    uint32_t D = 0, cc = hh;
                    while (cc != NIL && cc > w2) { D++; cc = prev[cc]; }
                    if (D >= B0) { verif += D; verif_calls++; }
                }
                if (!bound && (b != best || (b > prev_len && w2 != where))) mism++;
            }
            cur_at = where;
            cur_len = best < look ? best : look;
            if (cur_len == 3 && p - cur_at > 4096) cur_len = 2;
        }
        if (prev_len >= 3 && cur_len <= prev_len) {
            p += prev_len - 1;
            pending = 0; cur_len = 2;
        } else if (pending) { p++; } else { pending = 1; p++; }
        (void)prev_at;
    }
    printf("%s n=%u L%d: calls/byte %.3f  ref cand/byte %.1f (fresh %.1f lazy %.1f)  new cand/byte %.2f (fresh %.2f lazy %.2f; from chain top %.2f)  steps/call %.2f\n",
           argv[1], n, level, (double)calls / n, (double)ref_cand / n, (double)ref_fresh / n, (double)ref_lazy / n,
           (double)new_cand / n, (double)new_fresh / n, (double)new_lazy / n, (double)new_cand_top / n, (double)nsteps / calls);
    printf("   budget-bound calls %.4f of calls; verify calls %.4f, verify cand/byte %.2f; calls with <=1 candidate %.3f; unlimited != ref (unbound calls) %llu\n",
           (double)bound_calls / calls, (double)verif_calls / calls, (double)verif / n, (double)headonly / calls, (unsigned long long)mism);
    return 0;
}
