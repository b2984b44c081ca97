// ubench5.hip -- do MFMA accumulators in the AGPR half of the register file cost the VALU fewer cycles than in VGPRs?
//   per iteration and wave: 4 x v_mfma_scale_f32_32x32x64_f8f6f4 (C/D = 16 registers each, 4 accumulators) dealt between
//   4 x 16 VALU instructions (v_fma_f32 on independent registers); accumulators either compiler VGPRs or literal a[0:63].
// build: hipcc -O3 --offload-arch=gfx950 ubench5.hip -o ubench5
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define OP8(I) I(%0) I(%1) I(%2) I(%3) I(%4) I(%5) I(%6) I(%7)
#define I_FMA(r)  "v_fma_f32 " #r ", " #r ", %8, %9\n\t"
#define I_ADD(r)  "v_add_f32 " #r ", " #r ", %8\n\t"
#define I_EXP(r)  "v_exp_f32 " #r ", " #r "\n\t"
#define I_PK8(r)  "v_cvt_pk_fp8_f32 " #r ", " #r ", %8\n\t"
#define I_MAX3(r) "v_max3_i32 " #r ", " #r ", %8, %9\n\t"
#define I_CVT(r)  "v_cvt_f32_i32 " #r ", " #r "\n\t"
#define VALU16_OF(I) asm volatile(OP8(I) OP8(I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(ca), "v"(cb))
#define VALU16() do { if (OPK == 0) VALU16_OF(I_FMA); else if (OPK == 1) VALU16_OF(I_ADD); else if (OPK == 2) VALU16_OF(I_EXP); \
                      else if (OPK == 3) VALU16_OF(I_PK8); else if (OPK == 4) VALU16_OF(I_MAX3); else VALU16_OF(I_CVT); } while (0)

// MODE: 0 VALU only, 1 MFMA only (VGPR acc), 2 MFMA only (AGPR acc), 3 both (VGPR acc), 4 both (AGPR acc)
template <int MODE, int OPK>
__global__ void __launch_bounds__(256, 2) k(float *out, int iters)
{
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, ca = 0.999f, cb = 0.001f;
    v8i a = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (int)threadIdx.x};
    v16f o0 = {}, o1 = {}, o2 = {}, o3 = {};
    const int e8 = 0x7f7f7f7f;
    if (MODE == 2 || MODE == 4) {
        for (int i = 0; i < 64; i++) asm volatile("v_accvgpr_write_b32 a%0, 0" ::"n"(0));
        asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a48, 0" ::: "a0", "a16", "a32", "a48");
    }
    for (int it = 0; it < iters; it++) {
#define MF_V(acc) asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(e8))
#define MF_A(lo, hi) asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 a[" #lo ":" #hi "], %0, %1, a[" #lo ":" #hi "], %2, %2 op_sel_hi:[0,0,0]" :: "v"(a), "v"(b), "v"(e8) \
        : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
          "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63")
        if (MODE == 1 || MODE == 3) MF_V(o0);
        if (MODE == 2 || MODE == 4) MF_A(0, 15);
        if (MODE == 0 || MODE >= 3) VALU16();
        if (MODE == 1 || MODE == 3) MF_V(o1);
        if (MODE == 2 || MODE == 4) MF_A(16, 31);
        if (MODE == 0 || MODE >= 3) VALU16();
        if (MODE == 1 || MODE == 3) MF_V(o2);
        if (MODE == 2 || MODE == 4) MF_A(32, 47);
        if (MODE == 0 || MODE >= 3) VALU16();
        if (MODE == 1 || MODE == 3) MF_V(o3);
        if (MODE == 2 || MODE == 4) MF_A(48, 63);
        if (MODE == 0 || MODE >= 3) VALU16();
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    for (int i = 0; i < 16; i++) r += o0[i] + o1[i] + o2[i] + o3[i];
    if (MODE == 2 || MODE == 4) { float t; asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a5" : "=v"(t)); r += t; }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int OPK> static float run(float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, OPK>), dim3(512), dim3(256), 0, 0, out, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, OPK>), dim3(512), dim3(256), 0, 0, out, 4000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// ---- LDS fragment reads beside MFMAs / VALU: what does a ds_read_b128 (1 KiB into VGPRs) cost the SIMD? ----------------
// per iteration and wave: 4 x MFMA (or 64 v_fma) with NR ds_read (b128 or 2 x b64) dealt between them, lgkmcnt(0) at the end
template <int WORK, int NR, int W64>        // WORK: 0 none, 1 MFMA, 2 VALU (64 v_fma); NR reads per quarter; W64: use ds_read_b64 pairs
__global__ void __launch_bounds__(256, 2) kl(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char sm[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<int *>(sm)[i] = i;
    __syncthreads();
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, ca = 0.999f, cb = 0.001f;
    v8i a = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (int)threadIdx.x};
    v16f o0 = {}, o1 = {}, o2 = {}, o3 = {};
    const int e8 = 0x7f7f7f7f;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)sm + (threadIdx.x & 63) * 16;
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    typedef int v2i_ __attribute__((ext_vector_type(2)));
    v4i_ r0, r1, r2, r3; v2i_ h0, h1, h2, h3, h4, h5, h6, h7;
    int acc = 0;
    for (int it = 0; it < iters; it++) {
#define RD128(r, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(r) : "v"(addr))
#define RD64(r, off)  asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(r) : "v"(addr))
#define READS(q) do { if (NR >= 1) { if (W64) { RD64(h0, 0); RD64(h1, 8); } else RD128(r0, 0); } \
                      if (NR >= 2) { if (W64) { RD64(h2, 1024); RD64(h3, 1032); } else RD128(r1, 1024); } \
                      if (NR >= 3) { if (W64) { RD64(h4, 2048); RD64(h5, 2056); } else RD128(r2, 2048); } \
                      if (NR >= 4) { if (W64) { RD64(h6, 3072); RD64(h7, 3080); } else RD128(r3, 3072); } } while (0)
#define WORKQ(acc_) do { if (WORK == 1) asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc_) : "v"(a), "v"(b), "v"(e8)); \
                         if (WORK == 2) asm volatile(OP8(I_FMA) OP8(I_FMA) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(ca), "v"(cb)); } while (0)
        WORKQ(o0); READS(0); WORKQ(o1); READS(1); WORKQ(o2); READS(2); WORKQ(o3); READS(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (NR > 0) { if (W64) asm volatile("" :: "v"(h0), "v"(h1)); else asm volatile("" :: "v"(r0)); }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc;
    for (int i = 0; i < 16; i++) r += o0[i] + o1[i] + o2[i] + o3[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int WORK, int NR, int W64> static float runl(float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kl<WORK, NR, W64>), dim3(512), dim3(256), 0, 0, out, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kl<WORK, NR, W64>), dim3(512), dim3(256), 0, 0, out, 4000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int OPK> static void row(const char *name, float *out)
{
    const float v = run<0, OPK>(out), m = run<1, OPK>(out), b = run<3, OPK>(out);
    printf("%-22s VALU only %.3f   MFMA only %.3f   both %.3f   (sum %.3f, max %.3f: %2.0f%% of the smaller one hidden)\n", name, v, m, b, v + m, v > m ? v : m,
           100.0f * (v + m - b) / (v < m ? v : m));
}
int main()
{
    float *out; (void)hipMalloc(&out, 512 * 256 * 4);
    printf("# 2 waves/SIMD, 4000 iterations of 4 x (MX fp8 32x32x64 MFMA [64 cycles] + 16 VALU instructions of one kind); ms\n");
    row<0>("v_fma_f32", out); row<1>("v_add_f32", out); row<2>("v_exp_f32", out); row<3>("v_cvt_pk_fp8_f32", out);
    row<4>("v_max3_i32", out); row<5>("v_cvt_f32_i32", out);
    printf("accumulators in AGPRs (v_fma_f32): MFMA only %.3f  both %.3f\n", run<2, 0>(out), run<4, 0>(out));
    printf("# LDS reads (1 KiB per wave-instruction into VGPRs), 16 per iteration = per 4 MFMAs / 64 v_fma; ms (and cycles per read at 2.2 GHz)\n");
    { const float m0 = runl<1, 0, 0>(out), m4 = runl<1, 4, 0>(out), m2 = runl<1, 2, 0>(out), m4h = runl<1, 4, 1>(out);
      printf("MFMA:  no reads %.3f   8 x b128 %.3f   16 x b128 %.3f (%.1f cyc/read)   32 x b64 %.3f (%.1f cyc per KiB)\n", m0, m2, m4, (m4 - m0) * 2.2e6f / (4000 * 16 * 2), m4h, (m4h - m0) * 2.2e6f / (4000 * 16 * 2)); }
    { const float m0 = runl<2, 0, 0>(out), m4 = runl<2, 4, 0>(out), m4h = runl<2, 4, 1>(out);
      printf("VALU:  no reads %.3f   16 x b128 %.3f (%.1f cyc/read)   32 x b64 %.3f (%.1f cyc per KiB)\n", m0, m4, (m4 - m0) * 2.2e6f / (4000 * 16 * 2), m4h, (m4h - m0) * 2.2e6f / (4000 * 16 * 2)); }
    { const float m4 = runl<0, 4, 0>(out), m4h = runl<0, 4, 1>(out);
      printf("reads only: 16 x b128 %.3f (%.1f cyc/read)   32 x b64 %.3f\n", m4, m4 * 2.2e6f / (4000 * 16 * 2), m4h); }
    return 0;
}
