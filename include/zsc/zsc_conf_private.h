/*
 * zsc_conf_private.h -- assertion / warning / memory hooks for the MI355X build.
 *
 * Second of the two user-supplied configuration headers zsc requires
 * (reference README.md:28-33; example at test/zsc_test_private.h:39-90).
 * The host side of the drop-in library uses exactly these hooks, so that
 *   - NULL arguments die in ZSC_ASSERT (reference src/zsc_compress.c:56-59,
 *     src/zsc_uncompr.c:48-52; death tests test/zlib_gtest.cpp:2093-2396), and
 *   - diagnostics go through ZSC_WARN* and can be rerouted by an integrator.
 *
 * Define ZSC_QUIET_WARNINGS to compile the warnings out.
 */
#ifndef ZSC_CONF_PRIVATE_H
#define ZSC_CONF_PRIVATE_H

#include <assert.h>
#include <stdio.h>
#include <string.h>
#include "zsc/zsc_conf_global_types.h"

#ifndef ZSC_PRIVATE
#define ZSC_PRIVATE static
#endif

#define ZSC_ASSERT(c)                 assert(c)
#define ZSC_ASSERT1(c, a)             assert(c)
#define ZSC_ASSERT2(c, a, b)          assert(c)
#define ZSC_ASSERT3(c, a, b, d)       assert(c)
#define ZSC_ASSERT_DBL1(c, a)         assert(c)

#ifdef ZSC_QUIET_WARNINGS
#define ZSC_WARN(f)                   ((void)0)
#define ZSC_WARN1(f, a)               ((void)0)
#define ZSC_WARN2(f, a, b)            ((void)0)
#define ZSC_WARN3(f, a, b, c)         ((void)0)
#define ZSC_WARN4(f, a, b, c, d)      ((void)0)
#define ZSC_WARN5(f, a, b, c, d, e)   ((void)0)
#else
#define ZSC_WARN(f)                   fprintf(stderr, "ZSC WARNING " f "\n")
#define ZSC_WARN1(f, a)               fprintf(stderr, "ZSC WARNING " f "\n", a)
#define ZSC_WARN2(f, a, b)            fprintf(stderr, "ZSC WARNING " f "\n", a, b)
#define ZSC_WARN3(f, a, b, c)         fprintf(stderr, "ZSC WARNING " f "\n", a, b, c)
#define ZSC_WARN4(f, a, b, c, d)      fprintf(stderr, "ZSC WARNING " f "\n", a, b, c, d)
#define ZSC_WARN5(f, a, b, c, d, e)   fprintf(stderr, "ZSC WARNING " f "\n", a, b, c, d, e)
#endif

#define zmemcpy            memcpy
#define zmemcmp            memcmp
#define zmemzero(p, n)     memset((p), 0, (n))

#endif /* ZSC_CONF_PRIVATE_H */
