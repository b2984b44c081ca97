// sage_quant_math.h -- the INT8 rounding conventions of the reference's quantisers, shared by the stand-alone
// quantiser (sage_quant.hip) and the attention kernel's fused-Q prologue (sage_attn.hip) so both produce the same bits.
#pragma once
#include "sage_common.h"
#include "sage_kernels.h"

namespace sage {

// row (inside its block) -> scale group
__device__ __forceinline__ int group_of_row(int r, int gran, int warp)
{
    if (gran == GR_BLOCK) return 0;
    if (gran == GR_WARP) return r / warp;
    if (gran == GR_THREAD_Q) return (r / warp) * 8 + (r & 7);     // quant_per_thread.py:27-37
    return (r / warp) * 4 + ((r & 7) >> 1);                       // quant_per_thread.py:75-83
}

// scale of a quantisation group from its abs-max (QS_CUDA floors the abs-max at 1e-7 before calling, fused.cu:147)
__device__ __forceinline__ float quant_scale(float amax, int style)
{
    float sc = amax / 127.0f;
    if (style == QS_TRITON_THREAD) sc += 1e-7f;                  // quant_per_thread.py:41
    return sc;
}

// Triton convention: x / scale (correctly rounded IEEE quotient), +-0.5, truncate, clamp (quant_per_block.py:41-44).
// One IEEE reciprocal per group (`y`), then per element the FMA-based Markstein refinement q <- q + (x - scale*q) * y,
// twice: the first step makes q faithful, the second makes it the correctly rounded quotient (operands are far from
// overflow/underflow: |x| <= 127.5 * scale).  5 full-rate VALU ops instead of the compiler's ~10-instruction
// v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence per element.
__device__ __forceinline__ float quant_recip(float sc) { return (sc == 0.0f) ? 0.0f : 1.0f / sc; }

__device__ __forceinline__ int quant_round_triton(float x, float sc, float y)
{
    float t = x * y;
    t = __builtin_fmaf(__builtin_fmaf(-sc, t, x), y, t);
    t = __builtin_fmaf(__builtin_fmaf(-sc, t, x), y, t);
    t += (t >= 0.0f) ? 0.5f : -0.5f;
    int qi = (int)t;                                             // truncation toward zero
    qi = qi > 127 ? 127 : (qi < -128 ? -128 : qi);
    return (sc == 0.0f) ? 0 : qi;
}

// The same quotient when the scale cannot be zero or subnormal (QS_TRITON_THREAD: amax / 127 + 1e-7): |x / sc| < 127.001, so
// the clamp and the zero-scale select of quant_round_triton never act and are left out; +-0.5 is one v_bfi (copysign).
__device__ __forceinline__ int quant_round_triton_nz(float x, float sc, float y)
{
    float t = x * y;
    t = __builtin_fmaf(__builtin_fmaf(-sc, t, x), y, t);
    t = __builtin_fmaf(__builtin_fmaf(-sc, t, x), y, t);
    t += __builtin_copysignf(0.5f, t);
    return (int)t;
}

// Four elements of one scale group at a time, the same operations on 4-wide vectors: the compiler emits v_pk_mul_f32 / v_pk_fma_f32 and
// the four dependent chains (mul, 4 x fma, +-0.5, convert) advance in lock-step, two independent packed instructions per step, instead
// of one element's eight-instruction chain after the other (what it schedules for the scalar form when it is saving registers).
typedef float sage_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sage_f4 quant_quotient4(const sage_f4 x, float sc, float y)
{
    const sage_f4 ys = {y, y, y, y}, ns = {-sc, -sc, -sc, -sc};
    sage_f4 t = x * ys;
    t = __builtin_elementwise_fma(__builtin_elementwise_fma(ns, t, x), ys, t);
    t = __builtin_elementwise_fma(__builtin_elementwise_fma(ns, t, x), ys, t);
    return t;
}
__device__ __forceinline__ void quant_round_triton_nz4(const float (&x)[4], float sc, float y, int (&q)[4])
{
    sage_f4 t = quant_quotient4(sage_f4{x[0], x[1], x[2], x[3]}, sc, y);
    const sage_f4 h = {__builtin_copysignf(0.5f, t[0]), __builtin_copysignf(0.5f, t[1]), __builtin_copysignf(0.5f, t[2]), __builtin_copysignf(0.5f, t[3])};
    t += h;
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = (int)t[j];
}
__device__ __forceinline__ void quant_round_triton4(const float (&x)[4], float sc, float y, int (&q)[4])
{
    sage_f4 t = quant_quotient4(sage_f4{x[0], x[1], x[2], x[3]}, sc, y);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float u = t[j];
        u += (u >= 0.0f) ? 0.5f : -0.5f;
        int qi = (int)u;
        qi = qi > 127 ? 127 : (qi < -128 ? -128 : qi);
        q[j] = (sc == 0.0f) ? 0 : qi;
    }
}

// four INT8 lanes of one dword from four integers in [-128, 127]: three v_perm_b32 / v_or
__device__ __forceinline__ unsigned pack_int8x4(int q0, int q1, int q2, int q3)
{
    const unsigned lo = __builtin_amdgcn_perm((unsigned)q1, (unsigned)q0, 0x0c0c0400u);
    const unsigned hi = __builtin_amdgcn_perm((unsigned)q3, (unsigned)q2, 0x04000c0cu);
    return lo | hi;
}

// CUDA convention: x * (127 / amax), round to nearest even, saturate (cvt.rni.sat.s8.f32, fused.cu:164-172)
__device__ __forceinline__ int quant_round_cuda(float x, float inv)
{
    float t = __builtin_rintf(x * inv);
    t = fminf(fmaxf(t, -128.0f), 127.0f);
    return (int)t;
}

}  // namespace sage
