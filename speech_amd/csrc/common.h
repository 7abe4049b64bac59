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
