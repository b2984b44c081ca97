// ubench4.hip -- follow-up on the FP8 MFMA accumulator (ubench2 part C): how wide is the alignment window between the
// products of one MFMA, and how is the sum folded into C?   build: hipcc -O3 --offload-arch=gfx950 ubench4.hip -o ubench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void acc_k(const uint8_t *a64, const uint8_t *b64, float c, float *out, int scaled)
{
    const int lane = threadIdx.x & 63, g = lane >> 5;
    v8i av, bv;
    for (int w = 0; w < 8; w++) {
        uint32_t x = 0, y = 0;
        for (int j = 0; j < 4; j++) { x |= (uint32_t)a64[32 * g + 4 * w + j] << (8 * j); y |= (uint32_t)b64[32 * g + 4 * w + j] << (8 * j); }
        av[w] = (int)x; bv[w] = (int)y;
    }
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = c;
    if (scaled) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    else for (int s = 0; s < 4; s++) {
        const long a2 = (long)(((unsigned long)(unsigned)av[2 * s + 1] << 32) | (unsigned long)(unsigned)av[2 * s]);
        const long b2 = (long)(((unsigned long)(unsigned)bv[2 * s + 1] << 32) | (unsigned long)(unsigned)bv[2 * s]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2, b2, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = acc[0];
}
static uint8_t e4m3(float v)
{
    if (v == 0.0f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);
    int E = e - 1 + 7; float f = m * 2.0f - 1.0f;
    if (E <= 0) return s | (uint8_t)lrintf(v / ldexpf(1.0f, -9));
    return s | (uint8_t)(E << 3) | (uint8_t)lrintf(f * 8.0f);
}
static uint8_t *da, *db; static float *dout;
static float run(const float *a, const float *b, float c, int scaled)
{
    uint8_t ha[64], hb[64];
    for (int i = 0; i < 64; i++) { ha[i] = e4m3(a[i]); hb[i] = e4m3(b[i]); }
    (void)hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(acc_k, dim3(1), dim3(64), 0, 0, da, db, c, dout, scaled);
    float h; (void)hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    return h;
}
static void split(int k, float *a, float *b)       // a * b = 2^-k with both factors e4m3-representable
{
    if (k <= 9) { *a = ldexpf(1.0f, -k); *b = 1.0f; }
    else { *a = ldexpf(1.0f, -9); *b = ldexpf(1.0f, -(k - 9)); }
}
int main()
{
    (void)hipMalloc(&da, 64); (void)hipMalloc(&db, 64); (void)hipMalloc(&dout, 4);
    printf("# one product 1.0 at k=0 plus one product 2^-k at position pos (c = 0): result - 1 (exact: 2^-k), MX K=64 | 4 x K=16\n");
    const int poss[] = {1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63};
    for (int k = 6; k <= 18; k++) {
        printf("k=%2d exact %.3e :", k, ldexp(1.0, -k));
        for (int pi = 0; pi < 11; pi++) {
            float a[64] = {0}, b[64] = {0};
            a[0] = 1; b[0] = 1; split(k, &a[poss[pi]], &b[poss[pi]]);
            printf("  p%-2d %.2e|%.2e", poss[pi], run(a, b, 0.0f, 1) - 1.0f, run(a, b, 0.0f, 0) - 1.0f);
        }
        printf("\n");
    }
    printf("# c = 1.0 (no 1.0 product), a single product 2^-k at pos 0: result - 1\n");
    for (int k = 6; k <= 18; k++) {
        float a[64] = {0}, b[64] = {0};
        split(k, &a[0], &b[0]);
        printf("k=%2d exact %.3e : MX %.3e  K16 %.3e\n", k, ldexp(1.0, -k), run(a, b, 1.0f, 1) - 1.0f, run(a, b, 1.0f, 0) - 1.0f);
    }
    printf("# eight products 2^-k in ONE dword pair (pos 0..7) + c = 1: result - 1 (exact 8 * 2^-k)\n");
    for (int k = 18; k <= 18; k++) for (int cexp = 0; cexp <= 12; cexp += 2) {
        float a[64] = {0}, b[64] = {0};
        for (int i = 0; i < 8; i++) split(k, &a[i], &b[i]);
        const float c = ldexpf(1.0f, cexp);
        printf("c=2^%-2d k=%2d exact %.3e : MX %.3e  K16 %.3e   (ulp(c) = %.3e)\n", cexp, k, 8 * ldexp(1.0, -k), run(a, b, c, 1) - c, run(a, b, c, 0) - c, ldexp(1.0, cexp - 23));
    }
    printf("# rounding of the fold into C: c = 2^24 (ulp 2), products summing to r: result - c\n");
    for (int r = 1; r <= 7; r++) {
        float a[64] = {0}, b[64] = {0};
        for (int i = 0; i < r; i++) { a[i * 9 % 64] = 1.0f; b[i * 9 % 64] = 1.0f; }
        printf("sum=%d : MX %+.1f  K16 %+.1f   (RNE: %+.1f, truncate: %+.1f)\n", r, run(a, b, 16777216.0f, 1) - 16777216.0f, run(a, b, 16777216.0f, 0) - 16777216.0f,
               (double)((float)(16777216.0 + r)) - 16777216.0, floor(r / 2.0) * 2);
    }
    for (int r = 1; r <= 7; r++) {
        float a[64] = {0}, b[64] = {0};
        for (int i = 0; i < r; i++) { a[i * 9 % 64] = -1.0f; b[i * 9 % 64] = 1.0f; }
        printf("sum=-%d : MX %+.1f  K16 %+.1f\n", r, run(a, b, 33554432.0f, 1) - 33554432.0f, run(a, b, 33554432.0f, 0) - 33554432.0f);
    }
    return 0;
}
