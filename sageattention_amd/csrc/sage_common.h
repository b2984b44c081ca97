// sage_common.h -- device-side helpers shared by the gfx950 kernels.
//
// Everything here is CDNA4-specific (wave64, MFMA 32x32 fragment maps, LDS bank rules from
// /opt/skills/guides/MI355X_MICROARCH.md).  No CUDA compatibility layer, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sage {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int BLKQ = 128;   // query rows per workgroup (reference CTA_Q, qk_int_sv_f8_cuda_sm89.cuh:46)
constexpr int BLKK = 64;    // keys per KV tile        (reference CTA_K)
constexpr float kFp8Offset = 8.807f;   // log2(448): S_FP8_OFFSET, attn_utils.cuh:30
constexpr float kNegBig = -1.0e30f;

// dtype codes used across the C ABI
enum : int { DT_F16 = 0, DT_BF16 = 1 };

// ---- PV operand order ---------------------------------------------------------------------
// Inside one 64-key tile the K-dimension of the PV MFMA is fed in "position" order p = 0..63
// where position p holds token tau(p).  This is the order in which the swapped QK^T product
// (S^T = K Q^T, 32x32 MFMA C layout: row = (r&3) + 8*(r>>2) + 4*(lane>>5)) leaves the
// probabilities in a lane's registers, so P needs no cross-lane movement before PV:
//   p = 32*g + 8*c + j   (g = lane>>5, c = 16-key chunk 0..3, j = byte/elem in the operand)
//   tau = 16*c + 8*(j>>2) + 4*g + (j&3)
__host__ __device__ inline int pv_token_of_position(int p)
{
    const int g = p >> 5, c = (p >> 3) & 3, j = p & 7;
    return 16 * c + 8 * (j >> 2) + 4 * g + (j & 3);
}

// ---- LDS swizzles (conflict-free ds_read_b128 for "lane = row, fixed column chunk") --------
// Tile rows of 128 bytes (8 chunks of 16 B): chunk ^= (row >> 1) & 7
// Tile rows of  64 bytes (4 chunks of 16 B): chunk ^= (row >> 2) & 3
// Derived for the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32) and
// the 256-byte bank row: every group touches 16 distinct 16-B slots.
template <int ROW_BYTES>
__host__ __device__ inline int swz_chunk(int row, int chunk)
{
    if constexpr (ROW_BYTES == 128) return chunk ^ ((row >> 1) & 7);
    else if constexpr (ROW_BYTES == 64) return chunk ^ ((row >> 2) & 3);
    else return chunk;
}

// ---- scalar conversions --------------------------------------------------------------------
__device__ inline float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ inline uint16_t f32_to_bf16_rne(float f)
{
    uint32_t x = __float_as_uint(f);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}
__device__ inline float f16_to_f32(uint16_t h)
{
    _Float16 v; __builtin_memcpy(&v, &h, 2); return (float)v;
}
__device__ inline uint16_t f32_to_f16_rne(float f)
{
    _Float16 v = (_Float16)f; uint16_t h; __builtin_memcpy(&h, &v, 2); return h;
}
template <int DT> __device__ inline float ld16(uint16_t raw)
{
    if constexpr (DT == DT_F16) return f16_to_f32(raw); else return bf16_to_f32(raw);
}
template <int DT> __device__ inline uint16_t st16(float f)
{
    if constexpr (DT == DT_F16) return f32_to_f16_rne(f); else return f32_to_bf16_rne(f);
}

// max of a value with its lane^32 partner (the two lanes that share one query row)
__device__ inline float pair_max(float x)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float pair_sum(float x)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// packed (varlen) batches: rows [t0, t0 + len) of segment `seg` of a tensor of `total` rows -- a sequence (seg < nseq), or one of the two
// gaps outside every sequence that `k.mean(dim=0)` still averages over (core.py:432-434): nseq = the tail, nseq + 1 = the head
template <typename IntPtr>          // (a generic pointer, or a constant-address-space one for scalar loads)
__device__ __forceinline__ void varlen_segment(IntPtr cu, int nseq, int total, int seg, int &t0, int &len)
{
    if (seg < nseq) { t0 = cu[seg]; len = cu[seg + 1] - t0; }
    else if (seg == nseq) { t0 = cu[nseq]; len = total - t0; }
    else { t0 = 0; len = cu[0]; }
    len = len > 0 ? len : 0;
}

}  // namespace sage
