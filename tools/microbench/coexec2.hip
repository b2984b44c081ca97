// coexec2.hip -- how much independent VALU work hides under back-to-back MFMAs on gfx950?
// One stream per wave: 1 MFMA (i8 32x32x32, two alternating accumulators) followed by NV "softmax mix" VALU
// instructions (exp, fma, cvt_f32_i32, add on four independent registers), 3 waves/SIMD.
// Reports time vs NV and the pure-MFMA / pure-VALU times, and the same with MFMAs and VALU in separate phases
// (8 MFMAs, then 8*NV VALU) as the attention kernel's waves run them.
// build: hipcc -O3 --offload-arch=gfx950 coexec2.hip -o coexec2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MIX4 "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_i32 %2, %2\n v_add_f32 %3, %3, %4\n"

template <int NV4, int MODE>   // NV4 = groups of 4 VALU instrs per MFMA; MODE 0 interleaved, 1 phased, 2 VALU only, 3 MFMA only, 4 phased + staggered start
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    v16i c0 = {}, c1 = {};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    float fa = threadIdx.x, fb = fa + 1, fc = fa + 2, fd = fa + 3;
    const float x = 0.999f, y = 1e-3f;
    if (MODE == 4) {        // waves sharing a SIMD (blocks b, b+256, b+512..) start a fraction of a period apart
        const int slot = blockIdx.x / 256, nslot = gridDim.x / 256;
        const int pre = (8 * NV4 * slot) / nslot;
        for (int g = 0; g < pre; g++) asm volatile(MIX4 : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
    }
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NV4; g++) asm volatile(MIX4 : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
            }
        } else {
            if (MODE == 1 || MODE == 3 || MODE == 4) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                    else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                }
            }
            if (MODE == 1 || MODE == 4) asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n v_add_u32 %0, %0, %1" : "+v"(a[1]) : "v"(c0[0] + c1[0]));   // consume the MFMA results before the VALU phase
            if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
                for (int g = 0; g < 8 * NV4; g++) asm volatile(MIX4 : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
            }
        }
    }
    float s = fa + fb + fc + fd;
    for (int i = 0; i < 16; i++) s += (float)(c0[i] + c1[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)a[1];
}

template <int NV4, int MODE>
static float run(float *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV4, MODE>), dim3(blocks), dim3(256), 0, 0, out, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV4, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int NV4>
static void row(float *out, int blocks, int iters)
{
    const float ti = run<NV4, 0>(out, blocks, iters), tp = run<NV4, 1>(out, blocks, iters);
    const float tv = run<NV4, 2>(out, blocks, iters), tm = run<NV4, 3>(out, blocks, iters);
    const float ts = run<NV4, 4>(out, blocks, iters);
    printf("VALU/MFMA = %2d : interleaved %7.3f ms  phased %7.3f ms  staggered %7.3f ms | VALU only %7.3f  MFMA only %7.3f  sum %7.3f  max %7.3f\n",
           4 * NV4, ti, tp, ts, tv, tm, tv + tm, tv > tm ? tv : tm);
}

int main()
{
    float *out; (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
    for (int waves = 2; waves <= 4; waves++) {
        printf("== %d waves/SIMD (8 MFMA + 8*NV VALU per iteration, 1500 iterations)\n", waves);
        const int blocks = 256 * waves, iters = 1500;
        row<1>(out, blocks, iters);
        row<2>(out, blocks, iters);
        row<4>(out, blocks, iters);
        row<5>(out, blocks, iters);
    }
    return 0;
}
