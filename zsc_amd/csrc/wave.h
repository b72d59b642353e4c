/*
 * wave.h -- the two spellings of "one 64-lane wavefront" used by the kernels.
 *
 * Every kernel body in this directory is written once, as code for ONE gfx950
 * wavefront (64 lanes), in a style where
 *   - wave-uniform state (parse position, best length, chain budget ...) is an
 *     ordinary scalar,
 *   - per-lane state is declared with LANEVAR() and touched only inside a
 *     FOR_LANES { } region through LV(),
 *   - lanes talk to each other only through the collectives below, called from
 *     wave-uniform control flow.
 *
 * Built by hipcc (the product) FOR_LANES expands to nothing: the region is
 * ordinary SIMT code, LV(x) is the lane's register, the collectives are
 * v_cmp+s_mov ballots, v_readlane and DPP/LDS shuffles.
 *
 * Built with -DZSC_WAVE_EMU by g++ (tests/emu only, never shipped and never
 * linked into libzsc_hip.so) FOR_LANES is a loop over 64 lanes and LV(x) indexes
 * a 64-entry array, which lets the CPU test-suite execute the very same kernel
 * source lane by lane and compare every intermediate with the oracle.  This is
 * a test harness for kernel logic, not a fallback: the library has no code
 * path that reaches it.
 */
#ifndef ZSC_WAVE_H
#define ZSC_WAVE_H

#include <stdint.h>

#ifndef WAVE
#define WAVE 64 /* (the host emulation is also built with 16, to run group code at its real width) */
#endif

#ifdef ZSC_WAVE_EMU
/* ------------------------------------------------------------------ host */
#include <string.h>

#define DEV static inline
#define DEV_OUTLINED static /* rare paths kept out of the caller's register budget on the GPU */
#define GLOBAL_FN static
#define LDS_DECL(T, name, n) T name[n]
#define LANEVAR(T, name) T name[WAVE]
#define LV(name) name[_lane]
#define UNROLL_FULL
#define LANEARR(T, name, n) T name[n][WAVE] /* n values per lane, kept in registers on the GPU: constant indices only */
#define LVA(name, k) name[k][_lane]
#define LV_UNIFORM(name) name[0] /* a LANEVAR that holds the same value in every lane, read outside FOR_LANES */
#define FOR_LANES for (int _lane = 0; _lane < WAVE; ++_lane)
#define LANE (_lane)
#define ON_LANE0
#define WAVE_SYNC() ((void)0)
#define WG_BARRIER() ((void)0)

template <typename T>
static inline uint64_t emu_ballot(const T *p)
{
    uint64_t m = 0;
    for (int i = 0; i < WAVE; i++)
        if (p[i])
            m |= 1ull << i;
    return m;
}
#define BALLOT(name) emu_ballot(name)
#define READLANE(name, l) (name[(l)])

/* exclusive prefix sum over lanes; total returned */
template <typename T>
static inline T emu_exscan(const T *in, T *out)
{
    T run = 0;
    for (int i = 0; i < WAVE; i++) {
        T v = in[i];
        out[i] = run;
        run += v;
    }
    return run;
}
#define WAVE_EXSCAN(in, out, total) ((total) = emu_exscan(in, out))

template <typename T>
static inline T emu_sum(const T *in)
{
    T run = 0;
    for (int i = 0; i < WAVE; i++)
        run += in[i];
    return run;
}
#define WAVE_SUM(in) emu_sum(in)
template <typename T>
static inline T emu_min(const T *in)
{
    T m = in[0];
    for (int i = 1; i < WAVE; i++)
        m = in[i] < m ? in[i] : m;
    return m;
}
#define WAVE_MIN_U32(in) emu_min(in) /* the smallest value over the lanes (the same in every lane) */

static inline uint32_t ld_u32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint32_t ld_u16(const uint8_t *p)
{
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}
/* an unaligned 32-bit read from LDS: same as ld_u32 here */
static inline uint32_t lds_u32(const uint8_t *base, uint32_t idx)
{
    return ld_u32(base + idx);
}
#define LDS_ADD_U32(ptr, v) (*(ptr) += (v))
#define LDS_FETCH_ADD_U32(ptr, v) ((*(ptr) += (v)) - (v)) /* returns the old value */
#define LDS_OR_U32(ptr, v) (*(ptr) |= (v))
#define GLOBAL_OR_U32(ptr, v) (*(ptr) |= (v))
#define LDS_STORE_REL(ptr, v) (*(ptr) = (v))
#define LDS_LOAD_ACQ(ptr) (*(ptr))
#define CTZ64(x) __builtin_ctzll(x)
#define CTZ32(x) __builtin_ctz(x)
#define POPC64(x) __builtin_popcountll(x)
#define CLZ64(x) __builtin_clzll(x)
#define CLZ32(x) __builtin_clz(x)
static inline uint32_t emu_brev32(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}
#define BREV32(x) emu_brev32(x)
#define COPY16(dst, src) memcpy((dst), (src), 16)
#define RCP_F32(x) (1.0f / (x))
#define UNI(x) (x)
#define OPAQUE_UNI(x) ((void)0)

#else
/* ---------------------------------------------------------------- gfx950 */
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__
#define DEV_OUTLINED __device__ __noinline__
#define GLOBAL_FN __global__
#define LDS_DECL(T, name, n) __shared__ T name[n]
#define LANEVAR(T, name) T name
#define LV(name) name
#define UNROLL_FULL _Pragma("unroll")
#define LANEARR(T, name, n) T name[n]
#define LVA(name, k) name[k]
#define LV_UNIFORM(name) name
#define FOR_LANES
#define LANE ((int)(threadIdx.x & 63))
#define ON_LANE0 if ((threadIdx.x & 63) == 0)
/* Lanes of one wave exchange data through LDS without a hardware barrier (LDS
 * operations of a wave execute in order), but the COMPILER reasons per thread: a
 * load may legally be hoisted above a store that only another lane executes.
 * WAVE_SYNC() is the ordering point: a wavefront-scope release/acquire fence pair
 * around a wave barrier -- no instructions, only a scheduling/memory barrier. */
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#define WG_BARRIER() __syncthreads()

#define BALLOT(name) __ballot(name)
/* lane index must be wave-uniform: force it into an SGPR so this is v_readlane_b32 */
#define READLANE(name, l) \
    ((decltype(name))__builtin_amdgcn_readlane((int)(name), __builtin_amdgcn_readfirstlane((int)(l))))

/* Exclusive prefix sum over the 64 lanes on the DPP path of the vector ALU (no trips through LDS):
 * within each row of 16 by shifts of 1, 2, 4, 8 (a lane without a source in its row adds 0), then
 * lane 15 of a row onto the next row, and lane 31 onto the upper half. */
#define WAVE_DPP_ADD_STEP(x, ctrl, rows) \
    ((x) += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rows), 0xf, true))
DEV uint32_t wave_exscan_u32(uint32_t v, uint32_t &total)
{
    uint32_t x = v;
    WAVE_DPP_ADD_STEP(x, 0x111, 0xf); /* row_shr:1 */
    WAVE_DPP_ADD_STEP(x, 0x112, 0xf); /* row_shr:2 */
    WAVE_DPP_ADD_STEP(x, 0x114, 0xf); /* row_shr:4 */
    WAVE_DPP_ADD_STEP(x, 0x118, 0xf); /* row_shr:8 */
    WAVE_DPP_ADD_STEP(x, 0x142, 0xa); /* row_bcast:15 onto rows 1 and 3 */
    WAVE_DPP_ADD_STEP(x, 0x143, 0xc); /* row_bcast:31 onto rows 2 and 3 */
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}
#define WAVE_EXSCAN(in, out, total) ((out) = wave_exscan_u32((in), (total)))

DEV uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d);
    return v;
}
#define WAVE_SUM(in) wave_sum_u64(in)

/* The smallest value over the 64 lanes, on the data-parallel-primitive path of the vector ALU
 * (six v_min_u32 with a DPP operand and one v_readlane) instead of six trips through LDS
 * (ds_bpermute): within quads, within rows of 16, then lane 15 of a row into the next row and
 * lane 31 into the upper half, so that lane 63 holds the result. */
#define WAVE_DPP_MIN_STEP(v, ctrl, rows)                                                              \
    do {                                                                                              \
        const uint32_t _o = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rows), 0xf, false); \
        (v) = _o < (v) ? _o : (v);                                                                    \
    } while (0)
DEV uint32_t wave_min_u32(uint32_t v)
{
    WAVE_DPP_MIN_STEP(v, 0xb1, 0xf);  /* quad_perm [1,0,3,2] */
    WAVE_DPP_MIN_STEP(v, 0x4e, 0xf);  /* quad_perm [2,3,0,1] */
    WAVE_DPP_MIN_STEP(v, 0x141, 0xf); /* row_half_mirror */
    WAVE_DPP_MIN_STEP(v, 0x140, 0xf); /* row_mirror: every lane of a row holds the row's minimum */
    WAVE_DPP_MIN_STEP(v, 0x142, 0xa); /* row_bcast:15 into rows 1 and 3 */
    WAVE_DPP_MIN_STEP(v, 0x143, 0xc); /* row_bcast:31 into rows 2 and 3 */
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#define WAVE_MIN_U32(in) wave_min_u32(in)

DEV uint32_t ld_u32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
DEV uint32_t ld_u16(const uint8_t *p)
{
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}
/* An unaligned 32-bit read from LDS as two aligned dwords and a funnel shift.  gfx950 does
 * execute an unaligned ds_read_b32, but at ~47 LDS cycles a wave instead of a handful
 * (SQ_LDS_UNALIGNED_STALL: 153 of 207 CU cycles per input byte in the parser before this).
 * `base` must be 4-byte aligned and 4 readable bytes must follow base[idx + 3]. */
DEV uint32_t lds_u32(const uint8_t *base, uint32_t idx)
{
    const uint32_t *w = (const uint32_t *)(base + (idx & ~3u));
    return __builtin_amdgcn_alignbyte(w[1], w[0], idx & 3u);
}
#define LDS_ADD_U32(ptr, v) atomicAdd((ptr), (v))
#define LDS_FETCH_ADD_U32(ptr, v) atomicAdd((ptr), (v))
#define LDS_OR_U32(ptr, v) atomicOr((ptr), (v))
#define GLOBAL_OR_U32(ptr, v) atomicOr((ptr), (v))
/* a word in LDS through which one wave tells the others of its workgroup how far it has got:
 * what it stored to LDS before the release is there for whoever sees the new value */
#define LDS_STORE_REL(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_LOAD_ACQ(ptr) __hip_atomic_load((ptr), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define CTZ64(x) __builtin_ctzll(x)
#define CTZ32(x) __builtin_ctz(x)
#define POPC64(x) __builtin_popcountll(x)
#define CLZ64(x) __builtin_clzll(x)
#define CLZ32(x) __builtin_clz(x)
#define BREV32(x) __builtin_bitreverse32(x)
/* A value that is the same in every lane but was produced by a vector instruction
 * (an LDS or global load from a wave-uniform address): move it to an SGPR, so that
 * branches on it are scalar branches instead of exec-mask regions and arithmetic on
 * it runs on the scalar unit. */
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
/* A wave-uniform value the compiler may not reason about from here on (no instruction): what is
 * computed from it stays in the rarely taken branch it is needed in, instead of being hoisted out
 * of the loops around it and paid for on every trip. */
#define OPAQUE_UNI(x) asm volatile("" : "+s"(x))
/* both sides 16-byte aligned: one global_load_dwordx4 + one ds_write_b128 */
#define COPY16(dst, src) (*(uint4 *)(dst) = *(const uint4 *)(src))
#define RCP_F32(x) __builtin_amdgcn_rcpf(x) /* v_rcp_f32: 1 ulp */

#endif

#include "wave_group.h"

#endif /* ZSC_WAVE_H */
