/*
 * caller.c -- TEST INFRASTRUCTURE.  A plain C program that uses the one-shot API exactly as a
 * user of the reference does (reference test/zlib_gtest.cpp:390-431,728-755): it is compiled
 * against THE REFERENCE'S OWN HEADERS (/root/reference/include/zsc/zsc_pub.h,
 * zlib_types_pub.h) plus this repo's zsc_conf_* headers (which the reference asks its
 * integrator to supply), and linked against libzsc_hip.so.  If the two disagreed about a
 * prototype, an enum's width or gz_header's layout, this is where it would show.
 * tests/c_caller/Makefile builds it where /root/reference exists; the binary travels to the GPU
 * box and tests/test_gpu_c_caller.py runs it there.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zsc/zsc_pub.h"

static int fail(const char *what, int rc)
{
    printf("FAIL %s (rc %d)\n", what, rc);
    return 1;
}

int main(void)
{
    /* the reference's hello vector: zsc_compress("hello hello", level 6), SURVEY 8a13 */
    static const U8 hello[] = "hello hello";
    static const U8 want[] = {0x78, 0x9c, 0xcb, 0x48, 0xcd, 0xc9, 0xc9, 0x57, 0xc8, 0x00, 0x91, 0x00, 0x19, 0x91, 0x04, 0x49};
    U32 work_len = 0, unwork_len = 0, bound = 0;
    if (zsc_compress_get_min_work_buf_size(&work_len) != Z_OK)
        return fail("zsc_compress_get_min_work_buf_size", 0);
    if (zsc_uncompress_get_min_work_buf_size(&unwork_len) != Z_OK)
        return fail("zsc_uncompress_get_min_work_buf_size", 0);
    U8 *work = malloc(work_len), *unwork = malloc(unwork_len);
    U8 out[64];
    U32 out_len = sizeof out;
    ZlibReturn rc = zsc_compress(out, &out_len, hello, 11, 11, work, work_len, 6);
    if (rc != Z_OK || out_len != sizeof want || memcmp(out, want, sizeof want) != 0)
        return fail("hello vector", (int)rc);

    /* a 64 KiB buffer (BASELINE config 1) through zsc_compress and back, then as a gzip member
     * with a caller's header (gz_header crosses the boundary by layout) */
    const U32 n = 65536;
    U8 *src = malloc(n), *back = malloc(n);
    U32 x = 12345;
    for (U32 i = 0; i < n; i++) {
        x = x * 1103515245u + 12345u;
        src[i] = (U8)("the quick brown fox jumps over the lazy dog "[(x >> 16) % 44]);
    }
    if (zsc_compress_get_max_output_size(n, n, 6, &bound) != Z_OK)
        return fail("zsc_compress_get_max_output_size", 0);
    U8 *comp = malloc(bound);
    U32 comp_len = bound;
    rc = zsc_compress(comp, &comp_len, src, n, n, work, work_len, 6);
    if (rc != Z_OK)
        return fail("zsc_compress 64 KiB", (int)rc);
    U32 back_len = n, used = comp_len;
    rc = zsc_uncompress(back, &back_len, comp, &used, unwork, unwork_len);
    if (rc != Z_OK || back_len != n || used != comp_len || memcmp(back, src, n) != 0)
        return fail("zsc_uncompress 64 KiB", (int)rc);

    gz_header head;
    memset(&head, 0, sizeof head);
    U8 name[] = "caller.bin";
    head.name = name;
    head.name_max = sizeof name;
    head.os = 3;
    head.time = 1234567;
    U32 gz_bound = 0;
    if (zsc_compress_get_max_output_size_gzip(n, n, 9, &head, &gz_bound) != Z_OK)
        return fail("zsc_compress_get_max_output_size_gzip", 0);
    U8 *gz = malloc(gz_bound);
    U32 gz_len = gz_bound;
    rc = zsc_compress_gzip(gz, &gz_len, src, n, n, work, work_len, 9, &head);
    if (rc != Z_OK || gz[0] != 0x1f || gz[1] != 0x8b || !(gz[3] & 0x08))
        return fail("zsc_compress_gzip", (int)rc);
    gz_header got;
    memset(&got, 0, sizeof got);
    U8 got_name[32];
    got.name = got_name;
    got.name_max = sizeof got_name;
    back_len = n;
    used = gz_len;
    memset(back, 0, n);
    rc = zsc_uncompress_gzip(back, &back_len, gz, &used, unwork, unwork_len, &got);
    if (rc != Z_OK || back_len != n || memcmp(back, src, n) != 0 || strcmp((char *)got_name, "caller.bin") != 0 ||
        got.time != 1234567 || got.done != 1)
        return fail("zsc_uncompress_gzip", (int)rc);

    /* error convention: a destination that is too small */
    U32 small = 100;
    rc = zsc_compress(comp, &small, src, n, n, work, work_len, 6);
    if (rc != Z_BUF_ERROR)
        return fail("Z_BUF_ERROR expected", (int)rc);
    printf("OK hello=%u bytes, 64 KiB -> %u bytes (zlib), %u bytes (gzip with name)\n", out_len, comp_len, gz_len);
    return 0;
}
