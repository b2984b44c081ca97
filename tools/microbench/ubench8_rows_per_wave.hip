// ubench8: is the attention tile faster per (query, key) pair with 16 query rows per wave at 3-4 waves per SIMD than with 32 rows per wave at 2?
// Replays the steady loop's instruction mix (no memory traffic) per wave and iteration:
//   ROWS = 32: 32 x (bias add, scale fma, exp2, row-sum add) + 16 v_cvt_pk_fp8_f32 + 16 v_max3_i32 + 8 v_mfma_i32_32x32x32_i8 + 4 v_mfma_f32_32x32x64_f8f6f4
//   ROWS = 16: half of each VALU count + 8 v_mfma_i32_16x16x64_i8 + 4 v_mfma_f32_16x16x128_f8f6f4 (the same MACs per (query, key) pair)
// with the MFMAs dealt between the VALU groups.  Occupancy is set by the dynamic LDS size (160 KB per CU, 4 waves per workgroup = 1 per SIMD).
// Output: ns per 64-key tile of 32 query rows per SIMD (ROWS = 16: two wave-tiles).
// build: hipcc -O3 --offload-arch=gfx950 ubench8_rows_per_wave.hip -o ubench8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int ROWS>
__global__ void __launch_bounds__(256) tile_k(float *out, int iters)
{
    extern __shared__ unsigned char smem[];
    constexpr int NS = ROWS;                       // scores per lane and tile: 64 keys x ROWS rows / 64 lanes
    int s[NS];
    float e[NS];
    int pw[NS / 4];
    v4i ka = {(int)threadIdx.x, 1, 2, 3}, qb = {4, 5, 6, (int)threadIdx.x};
    v8i va = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    v16i c32[2] = {};
    v16f o32[4] = {};
    v4i c16[4] = {};
    v4f o16[4] = {};
#pragma unroll
    for (int i = 0; i < NS; i++) { s[i] = threadIdx.x * 3 + i; e[i] = 0; }
#pragma unroll
    for (int i = 0; i < NS / 4; i++) pw[i] = i;
    float cs = 1e-4f, mneg = -0.5f, rs0 = 0, rs1 = 0;
    int mx = 0;
    auto qk = [&](int u) {
        SB();
        if constexpr (ROWS == 32) c32[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ka, qb, c32[u & 1], 0, 0, 0);
        else c16[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qb, c16[u & 3], 0, 0, 0);
        SB();
    };
    auto pv = [&](int u) {
        v8i pb = {pw[0], pw[1], pw[2 % (NS / 4)], pw[3 % (NS / 4)], pw[0], pw[1], pw[2 % (NS / 4)], pw[3 % (NS / 4)]};
        SB();
        if constexpr (ROWS == 32) o32[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, pb, o32[u & 3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else o16[u & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, pb, o16[u & 3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        SB();
    };
    auto grp4 = [&](int i0) {       // four scores: bias add, scale fma, exp2, row sum, two fp8 packs
        float t0, t1, t2, t3;
        asm volatile("v_add_f32 %0, 0xbe22f983, %4\n\tv_add_f32 %1, 0xbe22f983, %5\n\tv_add_f32 %2, 0xbe22f983, %6\n\tv_add_f32 %3, 0xbe22f983, %7\n\t"
                     "v_fma_f32 %0, %0, %8, -%9\n\tv_fma_f32 %1, %1, %8, -%9\n\tv_fma_f32 %2, %2, %8, -%9\n\tv_fma_f32 %3, %3, %8, -%9\n\t"
                     "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(s[i0]), "v"(s[i0 + 1]), "v"(s[i0 + 2]), "v"(s[i0 + 3]), "v"(cs), "v"(mneg));
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                     "v_cvt_pk_fp8_f32 %6, %2, %3\n\tv_cvt_pk_fp8_f32 %6, %4, %5 op_sel:[0,0,1]"
                     : "+v"(rs0), "+v"(rs1) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(pw[i0 / 4]));
        e[i0] = t0;
    };
    for (int it = 0; it < iters; it++) {
        // row maximum: NS / 2 v_max3_i32
#pragma unroll
        for (int i = 0; i < NS; i += 2) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(mx) : "v"(s[i]), "v"(s[i + 1]));
        // NS / 4 groups of four scores with the tile's 12 MFMAs dealt between them
        constexpr int NG = NS / 4;
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            if constexpr (ROWS == 32) {
                if (gi < 4) pv(gi);
                else qk(gi - 4);           // 8 groups 4..7 -> one QK each, the other four below
                if (gi >= 4) qk(gi);
            } else {                       // 4 groups, 12 MFMAs: three per group
                pv(gi); qk(2 * gi); qk(2 * gi + 1);
            }
            grp4(4 * gi);
        }
        s[0] += (int)e[0] + c32[0][0] + c16[0][0];      // (keeps the chains alive; the asm statements are volatile, nothing is hoisted)
        mneg += 1e-9f * mx;
    }
    float acc = rs0 + rs1 + mneg + o32[0][0] + o32[1][1] + o32[2][2] + o32[3][3] + o16[0][0] + o16[1][1] + o16[2][2] + o16[3][3] + smem[threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int ROWS> static void run(int waves_per_simd, float *d)
{
    const int lds = 160 * 1024 / waves_per_simd - 1024;
    hipFuncSetAttribute((const void *)tile_k<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 4000, blocks = 256 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    tile_k<ROWS><<<blocks, 256, lds>>>(d, 100);
    hipEventRecord(a);
    tile_k<ROWS><<<blocks, 256, lds>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: waves_per_simd waves x iters wave-tiles of ROWS rows; normalise to tiles of 32 rows
    const double tiles32 = (double)waves_per_simd * iters * ROWS / 32.0;
    printf("ROWS=%2d  %d wave(s)/SIMD: %8.3f ms  -> %7.1f ns per 32-row x 64-key tile per SIMD\n", ROWS, waves_per_simd, ms, ms * 1e6 / tiles32);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
    run<32>(1, d); run<32>(2, d); run<32>(3, d);
    run<16>(2, d); run<16>(3, d); run<16>(4, d); run<16>(6, d);
    return 0;
}
