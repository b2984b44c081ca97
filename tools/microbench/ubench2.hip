// ubench2.hip -- round-2 microbenchmarks for the attention inner loop on gfx950.
//   A. issue cost of candidate VALU ops (wave64 cycles at 1..4 waves/SIMD)
//   B. cross-wave co-execution: one MFMA-only wave and one VALU-only wave on the SAME SIMD
//      (512-thread workgroup: waves w and w+4 share a SIMD), with and without s_nop spacing of the MFMAs
//   C. precision of the FP8 MFMA accumulator (SURVEY section 7 open question)
// build: hipcc -O3 --offload-arch=gfx950 ubench2.hip -o ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------------------------------- A
#define REP8(X) X X X X X X X X
#define BODY(INSTR) \
    asm volatile(REP8(INSTR) REP8(INSTR) REP8(INSTR) REP8(INSTR) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e2), "+v"(f2) : "v"(x), "v"(y), "v"(one));

template <int OP>
__global__ void __launch_bounds__(256) rate_k(float *out, int iters, float seed)
{
    float a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3, x = 0.999f, y = 1e-3f, one = 1.0f;
    v2f e2 = {a, b}, f2 = {c, d};
    for (int i = 0; i < iters; i++) {
        if (OP == 0) BODY("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n")
        if (OP == 1) BODY("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n")
        if (OP == 2) BODY("v_sub_f32 %0, %0, %6\n v_sub_f32 %1, %1, %6\n v_sub_f32 %2, %2, %6\n v_sub_f32 %3, %3, %6\n")
        if (OP == 3) BODY("v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %6\n")
        if (OP == 4) BODY("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n")
        if (OP == 5) BODY("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1\n")
        if (OP == 6) BODY("v_max_i32 %0, %0, %1\n v_max_i32 %1, %1, %2\n v_max_i32 %2, %2, %3\n v_max_i32 %3, %3, %0\n")
        if (OP == 7) BODY("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %1, %1, %2, %3\n v_max3_i32 %2, %2, %3, %0\n v_max3_i32 %3, %3, %0, %1\n")
        if (OP == 8) BODY("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n")
        if (OP == 9) BODY("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
        if (OP == 10) BODY("v_cvt_pk_fp8_f32 %0, %1, %2\n v_cvt_pk_fp8_f32 %1, %2, %3\n v_cvt_pk_fp8_f32 %2, %3, %0\n v_cvt_pk_fp8_f32 %3, %0, %1\n")
        if (OP == 11) BODY("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %8\n v_cvt_scalef32_pk_fp8_f32 %1, %2, %3, %8\n v_cvt_scalef32_pk_fp8_f32 %2, %3, %0, %8\n v_cvt_scalef32_pk_fp8_f32 %3, %0, %1, %8\n")
        if (OP == 12) BODY("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
        if (OP == 13) BODY("v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n")
        if (OP == 14) BODY("v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n")
        if (OP == 15) BODY("v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %5, %5, %4, %5\n v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %5, %5, %4, %5\n")
        if (OP == 16) BODY("v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n")
        if (OP == 17) BODY("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %1, %2, %3\n v_cvt_pk_f16_f32 %2, %3, %0\n v_cvt_pk_f16_f32 %3, %0, %1\n")
        if (OP == 18) BODY("v_cvt_scalef32_pk_fp8_f16 %0, %1, %8\n v_cvt_scalef32_pk_fp8_f16 %1, %2, %8\n v_cvt_scalef32_pk_fp8_f16 %2, %3, %8\n v_cvt_scalef32_pk_fp8_f16 %3, %0, %8\n")
        if (OP == 19) BODY("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n")
        if (OP == 20) BODY("v_pk_fma_f16 %0, %0, %6, %7\n v_pk_fma_f16 %1, %1, %6, %7\n v_pk_fma_f16 %2, %2, %6, %7\n v_pk_fma_f16 %3, %3, %6, %7\n")
        if (OP == 21) BODY("v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %1, %1, %2\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %3, %3, %0\n")
        if (OP == 22) BODY("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
        // the per-score chain of the FP8 softmax, as straight-line instruction mixes over independent registers
        if (OP == 30) BODY("v_cvt_f32_i32 %0, %0\n v_fma_f32 %1, %1, %6, %7\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %6\n")            // r1 mix
        if (OP == 31) BODY("v_sub_f32 %0, %0, %6\n v_fma_f32 %1, %1, %6, %7\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %6\n")              // magic-number mix
        if (OP == 32) BODY("v_sub_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %6\n")                  // all VOP2
        if (OP == 33) BODY("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n")      // 1 trans : 3 fma
        if (OP == 34) BODY("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n")                  // 1 trans : 3 add
        if (OP == 35) BODY("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %6\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %6\n")                      // 1 trans : 1 add
        if (OP == 36) BODY("v_cvt_pk_fp8_f32 %0, %1, %2\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n")       // 1 pack : 3 add
        if (OP == 37) BODY("v_exp_f32 %0, %0\n v_cvt_pk_fp8_f32 %1, %2, %3\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %6\n")               // exp + pack together
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e2[0] + e2[1] + f2[0] + f2[1];
}

static float g_clock_ghz = 2.4f;

template <int OP>
static int rate(const char *name, float *out, int waves)
{
    const int iters = 1500, blocks = 256 * waves;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_k<OP>, dim3(blocks), dim3(256), 0, 0, out, 200, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)waves * iters * 128.0;
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("  %-44s %7.3f ms  %6.3f ns/wave-instr/SIMD = %5.2f cyc @%.2f GHz\n", name, ms, ns_per, ns_per * g_clock_ghz, g_clock_ghz);
    return 0;
}

// ------------------------------------------------------------------------------------------- B
// 512-thread workgroup, 256 workgroups = one per CU: waves 0-3 run ROLE_A, waves 4-7 ROLE_B; wave w and w+4 share a SIMD.
// roles: 0 idle, 1 MFMA stream (i8 32x32x32, two accumulators), 2 VALU stream (softmax mix), 3 MFMA stream with s_nop spacing,
//        4 MX fp8 32x32x64 stream, 5 MX stream with s_nop spacing, 6 MFMA (i8) + 16 VALU interleaved, 7 MFMA phase (8) then VALU phase (128)
#define MIX4 "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_i32 %2, %2\n v_add_f32 %3, %3, %4\n"
template <int ROLE, int NOPS>
__device__ __forceinline__ void role_body(int iters, float *sink)
{
    v16i c0 = {}, c1 = {};
    v16f f0 = {}, f1 = {};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v8i a8 = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {4, 5, 6, (int)threadIdx.x, 1, 2, 3, 4};
    float fa = threadIdx.x, fb = fa + 1, fc = fa + 2, fd = fa + 3;
    const float x = 0.999f, y = 1e-3f;
    for (int i = 0; i < iters; i++) {
        if (ROLE == 1 || ROLE == 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                if (ROLE == 3) {
                    if (NOPS >= 1) asm volatile("s_nop 7");
                    if (NOPS >= 2) asm volatile("s_nop 7");
                    if (NOPS >= 3) asm volatile("s_nop 7");
                }
            }
            // 128 VALU-equivalents of time are not spent here: the MFMA wave does MFMAs only
        } else if (ROLE == 4 || ROLE == 5) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u & 1) f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                else f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                if (ROLE == 5) {
                    if (NOPS >= 1) asm volatile("s_nop 7\n s_nop 7");
                    if (NOPS >= 2) asm volatile("s_nop 7\n s_nop 7");
                    if (NOPS >= 3) asm volatile("s_nop 7\n s_nop 7");
                }
            }
        } else if (ROLE == 2) {
#pragma unroll
            for (int g = 0; g < 32; g++) asm volatile(MIX4 : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
        } else if (ROLE == 6) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; g++) asm volatile(MIX4 : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
            }
        } else if (ROLE == 7) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                if (NOPS >= 1) asm volatile("s_nop 7");
                if (NOPS >= 2) asm volatile("s_nop 7");
                if (NOPS >= 3) asm volatile("s_nop 7");
            }
            asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n v_add_u32 %0, %0, %1" : "+v"(a[1]) : "v"(c0[0] + c1[0]));
#pragma unroll
            for (int g = 0; g < 32; g++) asm volatile(MIX4 : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
        }
    }
    float s = fa + fb + fc + fd + (float)a[1];
    for (int i = 0; i < 16; i++) s += (float)(c0[i] + c1[i]) + f0[i] + f1[i];
    *sink = s;
}

template <int ROLE_A, int ROLE_B, int NOPS>
__global__ void __launch_bounds__(512) pair_k(float *out, int iters)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.0f;
    if (wave < 4) role_body<ROLE_A, NOPS>(iters, &s);
    else role_body<ROLE_B, NOPS>(iters, &s);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int ROLE_A, int ROLE_B, int NOPS>
static float pair_run(float *out, int blocks_per_cu)
{
    const int iters = 1500;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((pair_k<ROLE_A, ROLE_B, NOPS>), dim3(256 * blocks_per_cu), dim3(512), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pair_k<ROLE_A, ROLE_B, NOPS>), dim3(256 * blocks_per_cu), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// ------------------------------------------------------------------------------------------- C
// D = A(32x64 fp8) * B(64x32 fp8) + C with every row of A = (a_0 .. a_63) and every column of B = (b_0 .. b_63):
// all 1024 outputs equal sum_k a_k b_k + c.  Operand bytes: lane holds 32 consecutive k of its row/column (k = 32*(lane>>5) + j).
__global__ void acc_k(const uint8_t *a64, const uint8_t *b64, float c, float *out, int scaled)
{
    const int lane = threadIdx.x & 63, g = lane >> 5;
    v8i av, bv;
    for (int w = 0; w < 8; w++) {
        uint32_t x = 0, y = 0;
        for (int j = 0; j < 4; j++) {
            x |= (uint32_t)a64[32 * g + 4 * w + j] << (8 * j);
            y |= (uint32_t)b64[32 * g + 4 * w + j] << (8 * j);
        }
        av[w] = (int)x; bv[w] = (int)y;
    }
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = c;
    if (scaled) {
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
        for (int s = 0; s < 4; s++) {
            const long a2 = (long)(((unsigned long)(unsigned)av[2 * s + 1] << 32) | (unsigned long)(unsigned)av[2 * s]);
            const long b2 = (long)(((unsigned long)(unsigned)bv[2 * s + 1] << 32) | (unsigned long)(unsigned)bv[2 * s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2, b2, acc, 0, 0, 0);
        }
    }
    float mn = acc[0], mx = acc[0];
    for (int i = 1; i < 16; i++) { mn = fminf(mn, acc[i]); mx = fmaxf(mx, acc[i]); }
    out[2 * lane] = mn; out[2 * lane + 1] = mx;
}

static uint8_t e4m3(float v)      // exact encoder for powers of two and small integers used below
{
    if (v == 0.0f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);          // v = m 2^e, m in [0.5,1)
    int E = e - 1 + 7; float f = m * 2.0f - 1.0f;
    if (E <= 0) { int mant = (int)lrintf(v / ldexpf(1.0f, -9)); return s | (uint8_t)mant; }
    return s | (uint8_t)(E << 3) | (uint8_t)lrintf(f * 8.0f);
}

static int acc_case(const char *name, const float *a, const float *b, float c, double exact, uint8_t *da, uint8_t *db, float *dout)
{
    uint8_t ha[64], hb[64];
    for (int i = 0; i < 64; i++) { ha[i] = e4m3(a[i]); hb[i] = e4m3(b[i]); }
    CK(hipMemcpy(da, ha, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 64, hipMemcpyHostToDevice));
    for (int scaled = 1; scaled >= 0; scaled--) {
        hipLaunchKernelGGL(acc_k, dim3(1), dim3(64), 0, 0, da, db, c, dout, scaled);
        float h[128]; CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
        float mn = h[0], mx = h[1];
        for (int i = 0; i < 64; i++) { mn = fminf(mn, h[2 * i]); mx = fmaxf(mx, h[2 * i + 1]); }
        printf("  %-58s %s: got %.10g (min) %.10g (max)   exact %.10f  fp32(exact) %.10g\n", name, scaled ? "MX 32x32x64     " : "4 x 32x32x16 fp8", mn, mx, exact, (float)exact);
    }
    return 0;
}

int main()
{
    float *out; CK(hipMalloc(&out, 4096 * 512 * sizeof(float)));
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    if (clk > 0) g_clock_ghz = clk / 1e6f;

    printf("### C. FP8 MFMA accumulator precision\n");
    {
        uint8_t *da, *db; float *dout;
        CK(hipMalloc(&da, 64)); CK(hipMalloc(&db, 64)); CK(hipMalloc(&dout, 128 * sizeof(float)));
        float a[64], b[64];
        for (int i = 0; i < 64; i++) { a[i] = 0.125f; b[i] = 0.0625f; }
        a[0] = 256.0f; b[0] = 256.0f;
        acc_case("c=0, one product 2^16, 63 products 2^-7", a, b, 0.0f, 65536.0 + 63.0 / 128.0, da, db, dout);
        a[0] = 0.125f; b[0] = 0.0625f; a[63] = 256.0f; b[63] = 256.0f;
        acc_case("c=0, 63 products 2^-7, LAST product 2^16", a, b, 0.0f, 65536.0 + 63.0 / 128.0, da, db, dout);
        for (int i = 0; i < 64; i++) { a[i] = 1.0f; b[i] = 0.5f; }
        acc_case("c=2^24, 64 products 0.5 (each = half ulp of c)", a, b, 16777216.0f, 16777216.0 + 32.0, da, db, dout);
        for (int i = 0; i < 64; i++) { a[i] = 0.001953125f; b[i] = 0.001953125f; }     // 2^-9 * 2^-9 = 2^-18
        acc_case("c=256 (ulp 2^-15), 64 products 2^-18 (sum = 8 ulp)", a, b, 256.0f, 256.0 + 64.0 / 262144.0, da, db, dout);
        for (int i = 0; i < 64; i++) { a[i] = 0.0f; b[i] = 0.0f; }
        a[0] = 1.0f; b[0] = 1.0f; a[1] = 0.001953125f; b[1] = 0.015625f;  // 1 + 2^-15: needs 16 mantissa bits
        a[2] = 0.001953125f; b[2] = 0.001953125f;                        // + 2^-18
        a[3] = 0.001953125f; b[3] = 0.001953125f * 2;                    // + 2^-17
        acc_case("c=0, 1 + 2^-15 + 2^-18 + 2^-17", a, b, 0.0f, 1.0 + 1.0 / 32768 + 1.0 / 262144 + 1.0 / 131072, da, db, dout);
        a[0] = 448.0f; b[0] = 448.0f; a[1] = 0.001953125f; b[1] = 0.001953125f * 4;  // 200704 + 2^-16 (not representable: 2^17 ulp = 2^-6)
        a[2] = 1.0f; b[2] = 0.015625f; a[3] = 0; b[3] = 0;               // + 2^-6 exactly one ulp
        acc_case("c=0, 448*448 + 2^-6 + 2^-16", a, b, 0.0f, 200704.0 + 1.0 / 64 + 1.0 / 65536, da, db, dout);
        for (int i = 0; i < 64; i++) { a[i] = 1.5f; b[i] = (i & 1) ? -1.25f : 1.25f; }
        a[5] = 448.0f; b[5] = 0.001953125f;
        acc_case("c=1e-3, cancelling +-1.875 pairs + 448*2^-9", a, b, 1e-3f, 2.75 + (double)1e-3f, da, db, dout);
    }

    printf("### A. VALU issue cost (wave64 instructions, independent registers)\n");
    for (int waves = 1; waves <= 4; waves++) {
        printf("== %d wave(s) per SIMD\n", waves);
        rate<0>("v_fma_f32", out, waves);
        rate<1>("v_fmac_f32 (VOP2)", out, waves);
        rate<16>("v_add_f32", out, waves);
        rate<2>("v_sub_f32", out, waves);
        rate<3>("v_mul_f32", out, waves);
        rate<4>("v_max_f32", out, waves);
        rate<5>("v_max3_f32", out, waves);
        rate<6>("v_max_i32", out, waves);
        rate<7>("v_max3_i32", out, waves);
        rate<8>("v_cvt_f32_i32", out, waves);
        rate<9>("v_exp_f32", out, waves);
        rate<10>("v_cvt_pk_fp8_f32", out, waves);
        rate<11>("v_cvt_scalef32_pk_fp8_f32", out, waves);
        rate<17>("v_cvt_pk_f16_f32", out, waves);
        rate<18>("v_cvt_scalef32_pk_fp8_f16", out, waves);
        rate<19>("v_exp_f16", out, waves);
        rate<20>("v_pk_fma_f16", out, waves);
        rate<12>("v_mov_b32", out, waves);
        rate<13>("v_pk_add_f32", out, waves);
        rate<14>("v_pk_mul_f32", out, waves);
        rate<15>("v_pk_fma_f32", out, waves);
        rate<21>("v_ldexp_f32", out, waves);
        rate<22>("v_cndmask_b32", out, waves);
        rate<30>("mix cvt_f32_i32, fma, exp, add", out, waves);
        rate<31>("mix sub, fma, exp, add", out, waves);
        rate<32>("mix sub, mul, exp, add (all VOP2/VOP1)", out, waves);
        rate<33>("mix 1 exp : 3 fma", out, waves);
        rate<34>("mix 1 exp : 3 add", out, waves);
        rate<35>("mix 1 exp : 1 add", out, waves);
        rate<36>("mix 1 cvt_pk_fp8 : 3 add", out, waves);
        rate<37>("mix exp, cvt_pk_fp8, exp, add", out, waves);
    }

    printf("### B. one wave per role on each SIMD (512-thread workgroups, waves w / w+4 share a SIMD), 1500 iterations\n");
    printf("    per iteration: MFMA role = 8 x i8 32x32x32 (or 4 x MX fp8 32x32x64); VALU role = 128 softmax-mix instructions\n");
    for (int bpc = 1; bpc <= 2; bpc++) {
        printf("== %d workgroup(s) per CU = %d wave(s) of each role per SIMD\n", bpc, bpc);
        const float tm = pair_run<1, 0, 0>(out, bpc), tv = pair_run<0, 2, 0>(out, bpc), tb = pair_run<1, 2, 0>(out, bpc);
        printf("  i8 MFMA-only %.3f ms   VALU-only %.3f ms   both (different waves, same SIMD) %.3f ms   sum %.3f  max %.3f\n", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
        const float n1 = pair_run<3, 2, 1>(out, bpc), n2 = pair_run<3, 2, 2>(out, bpc), n3 = pair_run<3, 2, 3>(out, bpc);
        const float m1 = pair_run<3, 0, 1>(out, bpc), m2 = pair_run<3, 0, 2>(out, bpc), m3 = pair_run<3, 0, 3>(out, bpc);
        printf("  i8 MFMAs spaced by s_nop 7 x1/x2/x3: MFMA-only %.3f / %.3f / %.3f   both %.3f / %.3f / %.3f\n", m1, m2, m3, n1, n2, n3);
        const float xm = pair_run<4, 0, 0>(out, bpc), xb = pair_run<4, 2, 0>(out, bpc);
        const float x1 = pair_run<5, 2, 1>(out, bpc), x2 = pair_run<5, 2, 2>(out, bpc), x3 = pair_run<5, 2, 3>(out, bpc);
        printf("  MX fp8 MFMA-only %.3f ms   both %.3f ms   spaced by 2/4/6 s_nop 7: both %.3f / %.3f / %.3f\n", xm, xb, x1, x2, x3);
        const float va = pair_run<2, 2, 0>(out, bpc);
        printf("  VALU role on both waves %.3f ms (2x the VALU work)\n", va);
        const float ii = pair_run<6, 6, 0>(out, bpc), pp = pair_run<7, 7, 0>(out, bpc);
        const float p1 = pair_run<7, 7, 1>(out, bpc), p2 = pair_run<7, 7, 2>(out, bpc), p3 = pair_run<7, 7, 3>(out, bpc);
        printf("  both waves do MFMA+VALU: interleaved (1 MFMA : 16 VALU) %.3f   phased (8 MFMA, 128 VALU) %.3f   phased with s_nop x1/x2/x3 %.3f / %.3f / %.3f\n", ii, pp, p1, p2, p3);
    }
    return 0;
}
