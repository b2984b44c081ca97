// valu_rate.hip -- issue-rate microbenchmark for the VALU ops of the attention inner loop on gfx950.
// Each wave runs ITER x 32 independent instructions of one kind; 4 waves per SIMD.  Reports wave-instructions
// per ns per SIMD and the ratio to v_fma_f32 (a full-rate op: 4 cycles per wave64 instruction).
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY(INSTR) \
    asm volatile(REP8(INSTR) REP8(INSTR) REP8(INSTR) REP8(INSTR) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e2), "+v"(f2) : "v"(x), "v"(y));

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    float a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3, x = 0.999f, y = 1e-3f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f e2 = {a, b}, f2 = {c, d};
    for (int i = 0; i < iters; i++) {
        if (OP == 0) BODY("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n")
        if (OP == 1) BODY("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
        if (OP == 2) BODY("v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %5, %5, %4, %5\n v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %5, %5, %4, %5\n")
        if (OP == 3) BODY("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n")
        if (OP == 4) BODY("v_cvt_pk_fp8_f32 %0, %1, %2\n v_cvt_pk_fp8_f32 %1, %2, %3\n v_cvt_pk_fp8_f32 %2, %3, %0\n v_cvt_pk_fp8_f32 %3, %0, %1\n")
        if (OP == 5) BODY("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %1, %1, %2, %3\n v_max3_i32 %2, %2, %3, %0\n v_max3_i32 %3, %3, %0, %1\n")
        if (OP == 6) BODY("v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n")
        if (OP == 7) BODY("v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n")
        if (OP == 8) BODY("v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n")
        if (OP == 9) BODY("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n")       // 1 trans : 3 plain
        if (OP == 10) BODY("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n")
        if (OP == 11) BODY("v_cvt_f32_i32 %0, %0\n v_fma_f32 %1, %1, %6, %7\n v_exp_f32 %2, %2\n v_add_f32 %3, %3, %6\n")    // the softmax mix
        if (OP == 12) BODY("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
        if (OP == 13) BODY("v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %5, %5, %4, %5\n v_pk_fma_f32 %4, %4, %5, %4\n")  // trans + packed
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e2[0] + e2[1] + f2[0] + f2[1];
}

template <int OP>
static double run(const char *name, float *out, int blocks, double base)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) waves, each iters*128 instructions
    const double instr_per_simd = (double)blocks * 4 / 1024.0 * iters * 128.0;
    const double rate = instr_per_simd / (ms * 1e6);      // wave-instr per ns per SIMD
    printf("%-34s %8.3f ms  %7.4f wave-instr/ns/SIMD  %s%.2fx fma time\n", name, ms, rate, base > 0 ? "" : "(base) ", base > 0 ? base / rate : 1.0);
    return rate;
}

int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    for (int waves = 1; waves <= 4; waves += 3) {
        const int blocks = 256 * waves;                   // `waves` waves per SIMD
        printf("== %d wave(s) per SIMD\n", waves);
        double b = run<0>("v_fma_f32", out, blocks, 0);
        run<1>("v_exp_f32", out, blocks, b);
        run<10>("v_exp_f16", out, blocks, b);
        run<2>("v_pk_fma_f32", out, blocks, b);
        run<7>("v_pk_add_f32", out, blocks, b);
        run<8>("v_pk_mul_f32", out, blocks, b);
        run<3>("v_cvt_f32_i32", out, blocks, b);
        run<4>("v_cvt_pk_fp8_f32", out, blocks, b);
        run<5>("v_max3_i32", out, blocks, b);
        run<6>("v_add_f32", out, blocks, b);
        run<12>("v_cndmask_b32", out, blocks, b);
        run<9>("mix 1 exp : 3 fma", out, blocks, b);
        run<11>("mix cvt,fma,exp,add", out, blocks, b);
        run<13>("mix 1 exp : 3 pk_fma", out, blocks, b);
    }
    return 0;
}
