/*
 * zsc_conf_global_types.h -- platform type configuration for the MI355X build.
 *
 * zsc asks every integrator to supply this header (reference README.md:28-33;
 * the reference's own example lives at test/zsc_test_global_types.h:44-66).
 * This is the configuration our drop-in library is built with: C99 <stdint.h>
 * sized types, a compile-time assertion macro, and the two size-like typedefs
 * the public headers mention.
 */
#ifndef ZSC_CONF_GLOBAL_TYPES_H
#define ZSC_CONF_GLOBAL_TYPES_H

#include <stddef.h>
#include <stdint.h>

typedef uint8_t  U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef int32_t  I32;

#define U32_MAX ((U32)0xFFFFFFFFu)

/* negative array size => hard compile error when `cond` is false */
#define ZSC_COMPILE_ASSERT(cond, tag) typedef U8 (tag)[(cond) ? 1 : -1]

typedef U32    z_crc_t;   /* a CRC-32 value */
typedef size_t z_size_t;  /* largest object size the host can address */

#endif /* ZSC_CONF_GLOBAL_TYPES_H */
