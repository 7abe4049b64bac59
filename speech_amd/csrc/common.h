// common.h -- shared device helpers for libspeech_amd (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/speech_amd.h"

#define SA_WAVE 64

// Finite stand-in for log(0) in the log2-domain recurrences: differences of two SA_NEG values are 0 or a
// huge negative number, never NaN, so no -inf special-casing sits on the dependent chain.
#define SA_NEG (-1.0e30f)
#define SA_NEG_TEST (-1.0e29f)

#define SA_LOG2E 1.4426950408889634f
#define SA_LN2 0.6931471805599453f

// hipGetLastError() is per-thread sticky state shared with the host framework: clear what others left behind
// before launching, so that SA_CHECK_LAUNCH reports only this library's own launch failures.
#define SA_CLEAR_ERR() ((void)hipGetLastError())

#define SA_CHECK_LAUNCH()                                             \
    do {                                                              \
        hipError_t e__ = hipGetLastError();                           \
        if (e__ != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;    \
    } while (0)

static inline size_t sa_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ float sa_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float sa_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32

// The sticky health word of the persistent kernels (PersistHealth, gru.hip): err[0] is the device-side word the optimiser
// gates on; err[2 .. 3] hold the device address of a word in MAPPED HOST memory (or 0).  A failing block ORs its code into
// both -- the host sees a failure the moment it happens, and a healthy call costs the host nothing (rounds 1-3 copied the
// word to a pinned ring after every stack call: two 4-byte copies and ~20 us of launch gaps per train step).
__device__ __forceinline__ void sa_raise(unsigned* err, unsigned code) {
    atomicOr(err, code);
    // a GLOBAL-address-space pointer: through a generic one this is a flat_atomic_or, and a FLAT instruction that may be
    // pending makes hipcc's wait-count pass turn every later vmcnt wait into vmcnt(0) -- in the recurrence kernels that
    // was the top of every time step (round 5: the raise sits in their polling loops)
    typedef __attribute__((address_space(1))) unsigned sa_global_u32;
    sa_global_u32* hp = (sa_global_u32*)*reinterpret_cast<unsigned* const*>(err + 2);
    if (hp) __hip_atomic_fetch_or(hp, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Whole-wave shift by one lane (gfx9 DPP wave_shr / wave_shl): lane i receives lane i-1 (resp. i+1);
// the lane with no source keeps `edge`.
__device__ __forceinline__ float sa_wave_shr1(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float sa_wave_shl1(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// Full-wave reductions on the DPP network (no LDS traffic): 4 row_shr steps fold each 16-lane row into its
// lane 15, row_bcast15 / row_bcast31 fold the four rows into lane 63, v_readlane broadcasts.  ~7 VALU ops.
#define SA_DPP_F(old_, src_, ctrl_, rmask_)                                                                \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (old_)),                 \
                                                          __builtin_bit_cast(int, (src_)), (ctrl_), (rmask_), \
                                                          0xf, false))
__device__ __forceinline__ float sa_wave_max_dpp(float v) {
    v = fmaxf(v, SA_DPP_F(v, v, 0x111, 0xf));  // row_shr:1
    v = fmaxf(v, SA_DPP_F(v, v, 0x112, 0xf));  // row_shr:2
    v = fmaxf(v, SA_DPP_F(v, v, 0x114, 0xf));  // row_shr:4
    v = fmaxf(v, SA_DPP_F(v, v, 0x118, 0xf));  // row_shr:8
    v = fmaxf(v, SA_DPP_F(v, v, 0x142, 0xa));  // row_bcast:15 -> rows 1, 3
    v = fmaxf(v, SA_DPP_F(v, v, 0x143, 0xc));  // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float sa_wave_sum_dpp(float v) {
    v += SA_DPP_F(0.f, v, 0x111, 0xf);
    v += SA_DPP_F(0.f, v, 0x112, 0xf);
    v += SA_DPP_F(0.f, v, 0x114, 0xf);
    v += SA_DPP_F(0.f, v, 0x118, 0xf);
    v += SA_DPP_F(0.f, v, 0x142, 0xa);
    v += SA_DPP_F(0.f, v, 0x143, 0xc);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float sa_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sa_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- exact three-piece bf16 split of an fp32 number (the "x6" products of gemm_f32.hip and the fused recurrence kernels)
// a = a1 + a2 + a3 exactly, each piece the round-to-nearest-even bf16 of what the previous ones left.  A product of two
// split numbers is the sum of nine piece products, each exact in an fp32 accumulator; the six largest carry everything
// down to 2^-26 of the product (a quarter of an fp32 ulp).
typedef __bf16 sa_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned sa_cvt_pk_bf16(float lo, float hi) {  // {bf16(lo), bf16(hi)}
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sa_bf16_lo_as_f32(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float sa_bf16_hi_as_f32(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
// (x, y) -> three packed bf16 pairs whose sums are x and y
__device__ __forceinline__ void sa_split2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = sa_cvt_pk_bf16(x, y);
    const float rx = x - sa_bf16_lo_as_f32(p1), ry = y - sa_bf16_hi_as_f32(p1);  // exact
    p2 = sa_cvt_pk_bf16(rx, ry);
    const float sx = rx - sa_bf16_lo_as_f32(p2), sy = ry - sa_bf16_hi_as_f32(p2);  // exact
    p3 = sa_cvt_pk_bf16(sx, sy);
}
struct SaBf3 { sa_bf16x8 p[3]; };  // eight consecutive k of one row, as three bf16 planes
__device__ __forceinline__ SaBf3 sa_split8(float4 lo, float4 hi) {
    unsigned a[4], b[4], c[4];
    sa_split2(lo.x, lo.y, a[0], b[0], c[0]);
    sa_split2(lo.z, lo.w, a[1], b[1], c[1]);
    sa_split2(hi.x, hi.y, a[2], b[2], c[2]);
    sa_split2(hi.z, hi.w, a[3], b[3], c[3]);
    SaBf3 r;
    r.p[0] = __builtin_bit_cast(sa_bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
    r.p[1] = __builtin_bit_cast(sa_bf16x8, make_uint4(b[0], b[1], b[2], b[3]));
    r.p[2] = __builtin_bit_cast(sa_bf16x8, make_uint4(c[0], c[1], c[2], c[3]));
    return r;
}
// the six piece products in ascending size: (plane of A, plane of B)
#define SA_X6_PA(o) ((o) == 0 ? 2 : ((o) == 1 ? 0 : ((o) == 2 ? 1 : ((o) == 3 ? 0 : ((o) == 4 ? 1 : 0)))))
#define SA_X6_PB(o) ((o) == 0 ? 0 : ((o) == 1 ? 2 : ((o) == 2 ? 1 : ((o) == 3 ? 1 : ((o) == 4 ? 0 : 0)))))
