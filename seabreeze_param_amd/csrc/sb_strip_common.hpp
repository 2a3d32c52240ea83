// sb_strip_common.hpp -- what the two marching-strip contrast kernels share: k_strip (sb_strip_kernel.hip: search radii up
// to 16, 64-column rows, either precision) and k_strip32 (sb_strip32_kernel.hip: radii up to 31, 96-column rows, single
// precision).   ref: generic/sea_breeze_diag.f90:188-216
#pragma once
#include "sb_thc_common.hpp"

typedef unsigned long long u64;

// inclusive prefix sums over the 64 lanes of a wave of two 64-bit integers at once: per step and value one
// v_add_co_u32 + one v_addc_co_u32, the lane shift fused into the add (DPP).  The two chains alternate, so a value
// written by one step is read by the next four instructions later (a DPP read needs two wait states after a VALU
// write; hipcc pads nothing inside an asm statement -- hence also the leading s_nop).  Needs all 64 lanes active.
__device__ __forceinline__ void sb_scan2_u64(u64 &a, u64 &b) {
    unsigned al = (unsigned)a, ah = (unsigned)(a >> 32), bl = (unsigned)b, bh = (unsigned)(b >> 32);
#define SB_SCAN_STEP(ctl)                                          \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctl "\n\t"               \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctl "\n\t"         \
    "v_add_co_u32_dpp %2, vcc, %2, %2 " ctl "\n\t"               \
    "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctl "\n\t"
    asm volatile("s_nop 1\n\t"
                 SB_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 SB_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(al), "+v"(ah), "+v"(bl), "+v"(bh)::"vcc");
#undef SB_SCAN_STEP
    a = ((u64)ah << 32) | al;
    b = ((u64)bh << 32) | bl;
}

// ... of one 64-bit integer (rows that lie on one side of the coast: the land-side sums are all zero, or equal the
// sums over all cells); the steps follow each other directly, so each is padded to the two wait states of a DPP read
__device__ __forceinline__ void sb_scan1_u64(u64 &a) {
    unsigned al = (unsigned)a, ah = (unsigned)(a >> 32);
#define SB_SCAN_STEP1(ctl)                                         \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " ctl "\n\t"               \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctl "\n\ts_nop 0\n\t"
    asm volatile("s_nop 1\n\t"
                 SB_SCAN_STEP1("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP1("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP1("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP1("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 SB_SCAN_STEP1("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 SB_SCAN_STEP1("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(al), "+v"(ah)::"vcc");
#undef SB_SCAN_STEP1
    a = ((u64)ah << 32) | al;
}

__device__ __forceinline__ u64 sb_uniform64(u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// schedule entry of a step: position | flags
#define SCH_Q2 (1u << 16)          // the block two positions up is active: its band cells are queried in this step's S2
#define SCH_RESTART (1u << 17)     // the block before is not staged: the tables start afresh here
#define SCH_DRAIN (1u << 18)       // no block: the step behind the last block of a run, in which ...
#define SCH_Q1 (1u << 19)          // ... the block one position up (the run's last active one) is queried
#define SCH_IDLE (1u << 20)        // nothing but the loads of the block three steps on (warm-up and padding steps)
#define SCH_QI_SHIFT 22            // bits 22 .. 27: number of the step's cell list in the stored plan

typedef const __attribute__((address_space(4))) u64 *cu64p;          // read-only planes: scalar loads

// A cell whose window outgrows the tables (none on a grid whose distance field was made with a window of at most 15
// cells) is marked during the march -- a NaN of this payload in thc -- and handled after it, by the one copy of the
// global-memory search that the kernel holds (three inlined copies inside the march's loop tripled the code there, and
// a call would have the compiler wait for the prefetched blocks around it).
template <typename T> __device__ __forceinline__ T strip_mark();
template <> __device__ __forceinline__ double strip_mark<double>() { return __longlong_as_double(0x7ff85ea5b4ee2e00ll); }
template <> __device__ __forceinline__ float strip_mark<float>() { return __uint_as_float(0x7fc5ea5bu); }
__device__ __forceinline__ bool strip_is_mark(double v) { return __double_as_longlong(v) == 0x7ff85ea5b4ee2e00ll; }
__device__ __forceinline__ bool strip_is_mark(float v) { return __float_as_uint(v) == 0x7fc5ea5bu; }

