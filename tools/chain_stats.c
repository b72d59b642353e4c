/* chain_stats.c -- development aid: statistics of the reference's lazy parse
 * (deflate_slow + longest_match, src/deflate.c:1400-1518,1989-2122) on one file, to size
 * the GPU parser's work: candidates walked, pre-check passers, compares, per call.
 * Window sliding is approximated by "distance <= MAX_DIST" (statistics only). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define MAXD 32506u
static int cmpu(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int level = argc > 2 ? atoi(argv[2]) : 6;
    static const int cfgs[10][4] = {{0,0,0,0},{4,4,8,4},{4,5,16,8},{4,6,32,32},{4,4,16,16},{8,16,32,32},{8,16,128,128},{8,32,128,256},{32,128,258,1024},{32,258,258,4096}};
    uint32_t good = cfgs[level][0], lazy = cfgs[level][1], nicec = cfgs[level][2], chain = cfgs[level][3];
    fseek(f, 0, SEEK_END); uint32_t n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *in = malloc(n + 300); memset(in, 0, n + 300); fread(in, 1, n, f);
    uint32_t *head = calloc(32768, 4), *prev = calloc(n + 1, 4);
    for (int i = 0; i < 32768; i++) head[i] = 0xffffffffu;
    uint64_t iters = 0, calls = 0, calls2 = 0, callsL = 0, cand = 0, cand2 = 0, candL = 0, tri = 0, pass = 0, improve = 0, lcpbytes = 0, lcpcalls = 0;
    uint64_t maybe4 = 0, nsym = 0, nmatch = 0, passnoimp = 0, firstpass_tri=0;
    uint32_t *hist = malloc(4 * (n + 1)); uint32_t nh = 0;
    uint32_t *histtri = malloc(4 * (n + 1));
    uint32_t p = 0, cur_len = 2, cur_at = 0; int pending = 0;
#define INS(x) do { if ((x) + 2 < n) { uint32_t h = ((in[x] << 10) ^ (in[(x)+1] << 5) ^ in[(x)+2]) & 0x7fff; prev[x] = head[h]; head[h] = (x); } } while (0)
    while (p < n) {
        uint32_t look = n - p; iters++;
        uint32_t hh = 0xffffffffu;
        if (look >= 3) { uint32_t h = ((in[p] << 10) ^ (in[p+1] << 5) ^ in[p+2]) & 0x7fff; hh = head[h]; prev[p] = hh; head[h] = p; }
        uint32_t prev_len = cur_len, prev_at = cur_at; cur_len = 2;
        if (hh != 0xffffffffu && hh != 0 && prev_len < lazy && p - hh <= MAXD) {
            calls++; if (prev_len == 2) calls2++; else callsL++;
            uint32_t budget = chain, best = prev_len, nice = nicec, cap = look < 258 ? look : 258;
            if (prev_len >= good) budget >>= 2;
            if (nice > look) nice = look;
            uint32_t c = hh, nc = 0, ntri = 0;
            uint32_t limit = p > MAXD ? p - MAXD : 0;
            if (best < look) for (;;) {
                nc++;
                const uint8_t *m = in + c, *s = in + p;
                int t3 = m[0] == s[0] && m[1] == s[1] && m[2] == s[2];
                if (t3) { ntri++; if (m[3] == s[3]) maybe4++; }
                if (m[best] == s[best] && m[best-1] == s[best-1] && m[0] == s[0] && m[1] == s[1]) {
                    pass++; uint32_t len = 2; lcpcalls++;
                    while (len < cap && m[len] == s[len]) len++;
                    lcpbytes += len;
                    if (len > best) { cur_at = c; best = len; improve++; if (len >= nice) break; } else passnoimp++;
                    budget--;
                }
                c = prev[c];
                if (c == 0xffffffffu || c <= limit || budget == 0) break;
            }
            cand += nc; tri += ntri; if (prev_len == 2) cand2 += nc; else candL += nc;
            hist[nh] = nc; histtri[nh] = ntri; nh++;
            cur_len = best < look ? best : look;
            if (cur_len == 3 && p - cur_at > 4096) cur_len = 2;
        }
        if (prev_len >= 3 && cur_len <= prev_len) {
            nsym++; nmatch++;
            for (uint32_t k = prev_len - 2; k; k--) { p++; INS(p); }
            pending = 0; cur_len = 2; p++;
        } else if (pending) { nsym++; p++; } else { pending = 1; p++; }
    }
    qsort(hist, nh, 4, cmpu); qsort(histtri, nh, 4, cmpu);
    printf("%s n=%u L%d: iters/byte %.3f calls/byte %.3f (fresh %.3f lazy %.3f) cand/byte %.1f (fresh %.1f lazy %.1f) cand/call %.1f tri/call %.1f\n",
           argv[1], n, level, (double)iters / n, (double)calls / n, (double)calls2 / n, (double)callsL / n, (double)cand / n, (double)cand2 / n, (double)candL / n, (double)cand / calls, (double)tri / calls);
    printf("  pass/call %.2f improve/call %.2f passnoimp/call %.2f lcpbytes/lcp %.1f maybe4/call %.1f syms/byte %.3f matches %.3f\n",
           (double)pass / calls, (double)improve / calls, (double)passnoimp / calls, (double)lcpbytes / (lcpcalls ? lcpcalls : 1), (double)maybe4 / calls, (double)nsym / n, (double)nmatch / n);
    printf("  cand/call pct: 50%% %u 75%% %u 90%% %u 99%% %u max %u | tri: 50%% %u 75%% %u 90%% %u 99%% %u max %u\n", hist[nh/2], hist[nh*3/4], hist[nh*9/10], hist[(uint64_t)nh*99/100], hist[nh-1],
           histtri[nh/2], histtri[nh*3/4], histtri[nh*9/10], histtri[(uint64_t)nh*99/100], histtri[nh-1]);
    return 0;
}
