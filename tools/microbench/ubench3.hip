// ubench3.hip -- "tile model": the VALU + MFMA instruction stream of one 32-query x 64-key FP8 attention tile
// (no memory traffic), replayed in different ORDERS to price instruction mixing and MFMA placement on gfx950.
//   per tile and wave: 16 v_max3_i32, 32 x (int->float, scale FMA, exp2, row-sum add), 16 v_cvt_pk_fp8_f32,
//   64 fold FMAs, 8 x v_mfma_i32_32x32x32_i8 (QK^T) + 4 x v_mfma_scale_f32_32x32x64_f8f6f4 (PV)
// build: hipcc -O3 --offload-arch=gfx950 ubench3.hip -o ubench3
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define I_CVT(d, a)        asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(d) : "v"(a))
#define I_SUB(d, a, b)     asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define I_FMA(d, a, b, c)  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define I_FMAC(d, a, b)    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b))
#define I_MUL(d, a, b)     asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define I_EXP(d, a)        asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a))
#define I_ADD(d, a)        asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(a))
#define I_MAX3(d, a, b)    asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b))
#define I_MAXI(d, a)       asm volatile("v_max_i32 %0, %0, %1" : "+v"(d) : "v"(a))
#define I_PK_LO(d, a, b)   asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b))
#define I_PK_HI(d, a, b)   asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(d) : "v"(a), "v"(b))
#define SB() __builtin_amdgcn_sched_barrier(0)

// ORDER: 0 per-score chains back to back (cvt,fma,exp,add), 1 skewed software-pipelined chains, 2/3/4 batches of 8/16/32 per type
// MF: 0 no MFMA, 1 phased (8 QK MFMAs, softmax, then 4 x (PV MFMA, 16 fold FMAs)), 2 spread (MFMAs dealt between VALU batches)
// MAGIC: int->float by v_sub_f32 (bias trick) instead of v_cvt_f32_i32;  FOLD: 0 none, 1 v_fma, 2 v_fmac
template <int ORDER, int MF, int MAGIC, int FOLD>
__global__ void __launch_bounds__(256) tile_k(float *out, int iters)
{
    int s[32];
    float e[32], x[32], f[32];
    float o[64];
    int pw[8];
    v16i c0 = {}, c1 = {};
    v16f t0 = {}, t1 = {};
    v4i ka = {(int)threadIdx.x, 1, 2, 3}, qb = {4, 5, 6, (int)threadIdx.x};
    v8i va = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll
    for (int i = 0; i < 32; i++) { s[i] = threadIdx.x * 3 + i; e[i] = 0; x[i] = 0; f[i] = 0; }
#pragma unroll
    for (int i = 0; i < 64; i++) o[i] = threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 8; i++) pw[i] = i;
    float cs = 1e-4f, mneg = -0.5f, magic = 12582912.0f, alpha = 0.999f;
    float rs0 = 0, rs1 = 0, rs2 = 0, rs3 = 0;
    int mx = 0, mxb = 0, mxc = 0, mxd = 0;

    auto qk = [&](int u) {          // one QK^T MFMA, pinned in place
        SB();
        if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ka, qb, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ka, qb, c0, 0, 0, 0);
        SB();
    };
    auto pvm = [&](int u) {         // one PV MFMA (K = 64, MX fp8), pinned in place
        v8i pb = {pw[0], pw[1], pw[2], pw[3], pw[4], pw[5], pw[6], pw[7]};
        SB();
        if (u & 1) t1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, pb, t1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else t0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, pb, t0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        SB();
    };
    auto conv = [&](int i) { if (MAGIC) I_SUB(f[i], s[i], magic); else I_CVT(f[i], s[i]); };
    auto addrs = [&](int i) { if ((i & 3) == 0) I_ADD(rs0, e[i]); else if ((i & 3) == 1) I_ADD(rs1, e[i]); else if ((i & 3) == 2) I_ADD(rs2, e[i]); else I_ADD(rs3, e[i]); };
    auto pack4 = [&](int w) { I_PK_LO(pw[w], e[4 * w], e[4 * w + 1]); I_PK_HI(pw[w], e[4 * w + 2], e[4 * w + 3]); };
    auto fold = [&](int j) { if (FOLD == 1) I_FMA(o[j], o[j], alpha, e[j & 31]); else if (FOLD == 2) I_FMAC(o[j], alpha, e[j & 31]); };

    for (int it = 0; it < iters; it++) {
        if (MF == 1) { for (int u = 0; u < 8; u++) qk(u); }
        // row max over the raw scores (16 x max3) + a handful of scalar-ish ops
#pragma unroll
        for (int i = 0; i < 16; i++) {      // four independent chains, as the kernel's two sub-tiles x two scale groups
            if ((i & 3) == 0) I_MAX3(mx, s[2 * i], s[2 * i + 1]); else if ((i & 3) == 1) I_MAX3(mxb, s[2 * i], s[2 * i + 1]);
            else if ((i & 3) == 2) I_MAX3(mxc, s[2 * i], s[2 * i + 1]); else I_MAX3(mxd, s[2 * i], s[2 * i + 1]);
            if (MF == 2 && (i & 3) == 3) qk(i >> 2);
        }
        I_MAXI(mx, mxb); I_MAXI(mxc, mxd); I_MAXI(mx, mxc);
        { float t; I_CVT(t, mx); I_MUL(t, t, cs); I_ADD(t, mneg); I_EXP(alpha, t); }
        if (ORDER == 0) {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                conv(i); I_FMA(x[i], f[i], cs, mneg); I_EXP(e[i], x[i]); addrs(i);
                if ((i & 3) == 3) pack4(i >> 2);
                if (MF == 2 && (i & 7) == 7) qk(4 + (i >> 3));
            }
        } else if (ORDER == 1) {
#pragma unroll
            for (int i = 0; i < 32 + 3; i++) {
                if (i < 32) conv(i);
                if (i >= 1 && i - 1 < 32) I_FMA(x[i - 1], f[i - 1], cs, mneg);
                if (i >= 2 && i - 2 < 32) I_EXP(e[i - 2], x[i - 2]);
                if (i >= 3) { addrs(i - 3); if (((i - 3) & 3) == 3) pack4((i - 3) >> 2); }
                if (MF == 2 && (i & 7) == 7) qk(4 + (i >> 3));
            }
        } else {
            constexpr int B = ORDER == 2 ? 8 : ORDER == 3 ? 16 : 32;
#pragma unroll
            for (int b0 = 0; b0 < 32; b0 += B) {
#pragma unroll
                for (int i = b0; i < b0 + B; i++) conv(i);
#pragma unroll
                for (int i = b0; i < b0 + B; i++) I_FMA(x[i], f[i], cs, mneg);
                if (MF == 2) qk(4 + (b0 / B) % 4);
#pragma unroll
                for (int i = b0; i < b0 + B; i++) I_EXP(e[i], x[i]);
#pragma unroll
                for (int i = b0; i < b0 + B; i++) addrs(i);
                if (MF == 2 && B == 32) { qk(5); qk(6); qk(7); }
                if (MF == 2 && B == 16) qk(6 + b0 / 16);
#pragma unroll
                for (int w = b0 / 4; w < (b0 + B) / 4; w++) pack4(w);
            }
        }
        // PV + two-level fold
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            if (MF != 0) pvm(dt);
            if (FOLD != 0) {
#pragma unroll
                for (int j = 0; j < 16; j++) fold(dt * 16 + j);
            }
        }
        if (MF != 0) {      // consume the MFMA results so the chains stay live
            s[0] += c0[0] + c1[1];
            e[1] += t0[0] + t1[1];
        }
    }
    float r = rs0 + rs1 + rs2 + rs3 + alpha + (float)mx;
#pragma unroll
    for (int i = 0; i < 64; i++) r += o[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r += (float)pw[i];
#pragma unroll
    for (int i = 0; i < 16; i++) r += (float)(c0[i] + c1[i]) + t0[i] + t1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int ORDER, int MF, int MAGIC, int FOLD>
static float run(float *out, int waves)
{
    const int iters = 1000, blocks = 256 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((tile_k<ORDER, MF, MAGIC, FOLD>), dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((tile_k<ORDER, MF, MAGIC, FOLD>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (iters * waves);      // ns per tile per SIMD
}

template <int MF, int MAGIC, int FOLD>
static void row(const char *name, float *out, int waves)
{
    printf("  %-44s chains %6.0f  skewed %6.0f  batch8 %6.0f  batch16 %6.0f  batch32 %6.0f   ns/tile/SIMD\n", name,
           run<0, MF, MAGIC, FOLD>(out, waves), run<1, MF, MAGIC, FOLD>(out, waves), run<2, MF, MAGIC, FOLD>(out, waves),
           run<3, MF, MAGIC, FOLD>(out, waves), run<4, MF, MAGIC, FOLD>(out, waves));
}

int main()
{
    float *out; (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
    printf("# tile model: ns per (32 query x 64 key) wave-tile per SIMD; the round-1 kernel takes ~790 ns (1786 cycles at 2.26 GHz)\n");
    for (int waves = 1; waves <= 4; waves++) {
        printf("== %d wave(s) per SIMD\n", waves);
        row<0, 0, 0>("softmax VALU only, no fold", out, waves);
        row<0, 0, 1>("VALU only + fold (v_fma)", out, waves);
        row<0, 0, 2>("VALU only + fold (v_fmac)", out, waves);
        row<0, 1, 1>("VALU only, magic int->float, fold fma", out, waves);
        row<1, 0, 1>("MFMA phased + VALU + fold fma", out, waves);
        row<2, 0, 1>("MFMA spread + VALU + fold fma", out, waves);
        row<2, 1, 2>("MFMA spread, magic, fold fmac", out, waves);
        row<2, 1, 0>("MFMA spread, magic, no fold (alpha == 1)", out, waves);
    }
    return 0;
}
