/*
 * wave_group.h -- groups of lanes.  Included by wave.h, and again (it has no include guard) by
 * a kernel source that wants another group width for its own code:
 *     #undef ZSC_GROUP
 *     #define ZSC_GROUP 16
 *     #include "wave_group.h"      ... group code ...      and back to 64 the same way.
 *
 * Code whose control flow is serial per unit of work (one parse, one stream) spends most of
 * its instructions on that control flow; run by a whole wave those are scalar instructions for
 * ONE unit.  Written for a GROUP of lanes instead -- group-uniform values in vector registers,
 * identical in the group's lanes -- the units of one wave share every instruction they execute
 * at the same time: the lanes of a group vote (a slice of the wave's ballot) and read each
 * other's registers (ds_bpermute within the group); the groups of a wave diverge like any
 * threads do.  That pays when the groups mostly stay in the same piece of code (the inflate
 * symbol loop); it does not when they do not (four 16-lane LZ77 parsers per wave execute 326
 * instructions per input byte where one 64-lane parser executes 240: DESIGN.md section 5).
 *
 * On the host (tests/emu) a group is always the whole emulated wave; building that emulation
 * with 16-lane waves runs group code at the width it has on the GPU.
 */
#undef GRP
#undef GROUPS_PER_WAVE
#undef GGROUP
#undef GLANE
#undef FOR_GLANES
#undef ON_GLANE0
#undef GBALLOT
#undef GREADLANE
#undef GUNI
#undef GSUM64
#undef GMIN_U32
#undef GOPAQUE

#ifndef ZSC_GROUP
#define ZSC_GROUP 64
#endif

#if defined(ZSC_WAVE_EMU) || ZSC_GROUP == 64
#define GRP WAVE
#define GROUPS_PER_WAVE 1
#define GGROUP 0
#define GLANE LANE
#define FOR_GLANES FOR_LANES
#define ON_GLANE0 ON_LANE0
#define GBALLOT(name) BALLOT(name)
#define GREADLANE(name, l) READLANE(name, l)
#define GUNI(x) UNI(x)
#define GSUM64(in) WAVE_SUM(in)
#define GMIN_U32(in) WAVE_MIN_U32(in)
#define GOPAQUE(x) OPAQUE_UNI(x)
#else
#define GRP ZSC_GROUP
#define GROUPS_PER_WAVE (64 / ZSC_GROUP)
#define GGROUP ((int)((threadIdx.x & 63) / ZSC_GROUP))
#define GLANE ((int)(threadIdx.x & (ZSC_GROUP - 1)))
#define FOR_GLANES
#define ON_GLANE0 if ((threadIdx.x & (ZSC_GROUP - 1)) == 0)
#define GBALLOT(name) ((uint64_t)((__ballot(name) >> (threadIdx.x & (64 - ZSC_GROUP))) & ((1ull << ZSC_GROUP) - 1ull)))
#define GREADLANE(name, l) ((decltype(name))__shfl((int)(name), (int)(threadIdx.x & (64 - ZSC_GROUP)) + (int)(l)))
#define GUNI(x) (x)
#define GSUM64(in) group_sum_u64(in)
#define GMIN_U32(in) group_min_u32(in)
#define GOPAQUE(x) asm volatile("" : "+v"(x)) /* (group-uniform values live in vector registers) */
#ifndef ZSC_GROUP_SUM_DEFINED
#define ZSC_GROUP_SUM_DEFINED
DEV uint64_t group_sum_u64(uint64_t v)
{
#pragma unroll
    for (int d = ZSC_GROUP / 2; d >= 1; d >>= 1) /* (every group code in one source uses the same width) */
        v += __shfl_xor(v, d);
    return v;
}
DEV uint32_t group_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = ZSC_GROUP / 2; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
        v = o < v ? o : v;
    }
    return v;
}
#endif
#endif
