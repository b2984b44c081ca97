// coexec.hip -- do VALU instructions and MFMAs of the same SIMD overlap on gfx950?
//  A: MFMA-only waves            B: VALU-only waves         C: half the waves MFMA, half VALU (different waves)
//  D: one stream, 1 MFMA followed by 8 independent VALU ops (same wave)
// build: hipcc -O3 --offload-arch=gfx950 coexec.hip -o coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VALU8(INS) INS INS INS INS INS INS INS INS

__device__ __forceinline__ void mfma_loop(int iters, v16i &c0, v16i &c1, v4i a, v4i b)
{
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void valu_loop(int iters, float &a, float &b, float &c, float &d, float x, float y, int kind)
{
    for (int i = 0; i < iters; i++) {
        if (kind == 0)
            asm volatile(VALU8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         VALU8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
        else
            asm volatile(VALU8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_i32 %2, %2\n v_add_f32 %3, %3, %4\n")
                         VALU8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_i32 %2, %2\n v_add_f32 %3, %3, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
    }
}

// mode 0: all MFMA; 1: all VALU; 2: block parity decides; 3: same-wave interleave
template <int MODE, int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    v16i c0 = {}, c1 = {};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    float fa = threadIdx.x, fb = fa + 1, fc = fa + 2, fd = fa + 3;
    const float x = 0.999f, y = 1e-3f;
    if (MODE == 0 || (MODE == 2 && (blockIdx.x & 1) == 0)) mfma_loop(iters, c0, c1, a, b);
    else if (MODE == 1 || MODE == 2) valu_loop(iters, fa, fb, fc, fd, x, y, KIND);
    else {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
                else asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_i32 %2, %2\n v_add_f32 %3, %3, %4\n" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
                else asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_i32 %2, %2\n v_add_f32 %3, %3, %4\n" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(x), "v"(y));
            }
        }
    }
    float s = fa + fb + fc + fd;
    for (int i = 0; i < 16; i++) s += (float)(c0[i] + c1[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int KIND>
static float run(const char *name, float *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(256), 0, 0, out, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.3f ms\n", name, ms);
    return ms;
}

int main()
{
    float *out; (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 4000;
    // MFMA loop: 16 MFMA / iter; VALU loop: 64 instr / iter; interleave: 16 MFMA + 64 VALU / iter
    printf("-- 2 waves/SIMD of each kind where applicable (blocks = 512 single-kind, 1024 mixed)\n");
    float tm = run<0, 0>("A  MFMA only, 2 waves/SIMD (16 MFMA/iter)", out, 512, iters);
    float tv0 = run<1, 0>("B0 VALU fma only, 2 waves/SIMD (64 instr/iter)", out, 512, iters);
    float tv1 = run<1, 1>("B1 VALU softmax mix only, 2 waves/SIMD", out, 512, iters);
    float tc0 = run<2, 0>("C0 2 MFMA waves + 2 fma waves per SIMD", out, 1024, iters);
    float tc1 = run<2, 1>("C1 2 MFMA waves + 2 softmax-mix waves per SIMD", out, 1024, iters);
    float td0 = run<3, 0>("D0 same wave: MFMA + 4 fma interleaved, 2 waves/SIMD", out, 512, iters);
    float td1 = run<3, 1>("D1 same wave: MFMA + 4 mix interleaved, 2 waves/SIMD", out, 512, iters);
    float td0_4 = run<3, 0>("D0 same, 4 waves/SIMD (2x work)", out, 1024, iters);
    printf("overlap C0: %.2f (1 = perfect, 0 = serial)\n", (tm + tv0 - tc0) / (tm < tv0 ? tm : tv0));
    printf("overlap C1: %.2f\n", (tm + tv1 - tc1) / (tm < tv1 ? tm : tv1));
    printf("overlap D0: %.2f   D1: %.2f  D0x4waves time/2 = %.3f\n", (tm + tv0 - td0) / (tm < tv0 ? tm : tv0), (tm + tv1 - td1) / (tm < tv1 ? tm : tv1), td0_4 / 2);
    return 0;
}
