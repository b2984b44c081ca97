// sage_attn64.hip -- the FP8-PV attention kernel in its one-wave-per-SIMD form (round 3).
//
// Same arithmetic as sage_attn.hip's FP8 path (reference: csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:46-704,
// qk_int_sv_f8_cuda_sm90.cu:127-567): INT8 S^T = K Q^T on v_mfma_i32_32x32x32_i8, online softmax in the log2 domain with
// the 8.807 offset, P -> e4m3 in registers, O^T += V^T P^T on the K = 64 FP8 MFMA, FP32 accumulation through the MFMA's C
// operand.  What changes is the shape of the work:
//
//  * a workgroup = 4 waves = 256 query rows of one (batch, q-head); a wave owns 64 rows = two 32-row blocks rb = 0, 1 and
//    the whole 512-register file of its SIMD (one wave per SIMD): every K / V fragment a wave reads from LDS feeds twice
//    the matrix work, and every LDS-DMA'd tile serves 256 rows instead of 128.
//  * the register file is laid out by hand.  hipcc's allocator cannot pack fourteen 8- and 16-wide tuples (208 of 256
//    VGPRs) next to its own temporaries: with the tuples as C++ values -- free or pinned with physical-register constraints
//    -- it moved whole O tiles to the accumulator file and back around every MFMA.  So the kernel is compiled with
//    amdgpu_num_vgpr(48): the compiler owns v0..v47 (and a0..a47 as its spill space) and every other register is named
//    literally in the asm text:
//        v[48:63]   P  (2 row blocks x 8 words)              a[48:79]    Q fragments (2 row blocks x 4 k-steps x 4)
//        v[64:127]  S  (2 key halves x 2 row blocks x 16)     a[80:111]   K fragments (2 key halves x 4 k-steps x 4)
//        v[128:255] O  (4 channel tiles x 2 row blocks x 16)  a[112:143]  V fragments (4 channel tiles x 8)
//                                                             a[144:152]  lane constants (LDS / DMA offsets)
//    O, S and P stay in the architected VGPRs because the VALU works on them; the MFMA-only operands live in the
//    accumulator file (ds_read_b128 writes AGPRs directly, MFMA A / B operands may be AGPRs).
//  * half-rotated software pipeline.  One tile t is two phases:
//        A_t : VALU softmax(t, rb 0)   | MFMA  PV(t-1, rb 1)  QK^T(t, rb 1)    | LDS reads V(t), K(t+1)
//        B_t : VALU softmax(t, rb 1)   | MFMA  PV(t,   rb 0)  QK^T(t+1, rb 0)
//    so a phase's matrix work never depends on the VALU work beside it, the score tile of a row block is rewritten in
//    place by the next tile's QK^T as soon as its softmax has consumed it (64 score registers, not 128), and P needs no
//    second buffer.
//  * 4-slot LDS ring, LDS-DMA three tiles ahead, one counted `s_waitcnt vmcnt` + one `s_barrier` per tile.
//  * masked tiles (causal diagonal, ragged last tile) run the same phases: the masked scores are first overwritten in
//    place with the bit pattern of -FLT_MAX, and their probabilities are forced to exactly zero; a wave that has passed
//    its last visible tile only keeps the ring going.
//
// Hazards the text handles itself (hipcc pads nothing inside or around an asm statement, cdna_hip_programming.md 5.7):
//   MFMA result -> VALU reader: S tiles are read a phase after their last QK^T MFMA (>= 40 instructions), O tiles are
//   rescaled a phase after their PV MFMA; drains end in s_nop blocks.  VALU-written operand -> MFMA: P words are written
//   a phase before their PV (the first PV of phase B opens with s_nop 1).  v_exp_f32 -> reader: three instructions apart.
//   ds_read -> MFMA operand: explicit s_waitcnt lgkmcnt.  A fragment register is overwritten (ds_read) only after the last
//   MFMA that reads it has been issued, at least six instructions earlier.
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_quant_math.h"
#include <climits>
#include <type_traits>

#ifndef SAGE64_ABL     // timing ablations (wrong results): 1 no loop barrier, 2 no O rescale, 4 no loop LDS-DMA, 8 no fragment reads,
#define SAGE64_ABL 0   // 16 no MFMAs, 32 no softmax VALU, 64 no K-scale loads
#endif

namespace sage {
namespace a64 {

constexpr int BQ = 256;        // query rows per workgroup
constexpr int NSLOT = 4;       // LDS ring depth (tiles)
constexpr int AHEAD = 3;       // LDS-DMA distance (tiles)
constexpr float kSUnit = 67108864.0f;   // 2^26: the biased score registers count in units of 2^-26 (sage_attn.hip, kSInit / sfl)
#define SAGE64_NCOMP 48        // architected VGPRs (and AGPRs) left to the compiler

template <int D> struct Cfg {
    static constexpr int K_BYTES = BLKK * D;          // int8 K tile
    static constexpr int V_BYTES = D * 64;            // fp8 V^T image
    static constexpr int STAGE = K_BYTES + V_BYTES;
    static constexpr int O_BYTES = BQ * D * 2;
    static constexpr int LDS = (NSLOT * STAGE > O_BYTES) ? NSLOT * STAGE : O_BYTES;
    static constexpr int KSTEPS = D / 32;
    static constexpr int DT = D / 32;
    static constexpr int PCS = K_BYTES / 4096;        // 1-KiB LDS-DMA pieces per wave and image (4 waves)
};

// ---- register map (sized for D = 128) ------------------------------------------------------------------------------------------
constexpr int PV_(int rb, int w) { return 48 + rb * 8 + w; }                       // P word
constexpr int SV(int sb, int rb, int i) { return 64 + (sb * 2 + rb) * 16 + i; }    // S^T register
constexpr int OV(int dt, int rb, int i) { return 128 + (dt * 2 + rb) * 16 + i; }   // O^T register
constexpr int QA(int rb, int kk) { return 48 + (rb * 4 + kk) * 4; }
constexpr int KA(int sb, int kk) { return 80 + (sb * 4 + kk) * 4; }
constexpr int VA(int dt) { return 112 + dt * 8; }
constexpr int SA_KADDR = 144;  // a144..a147: LDS address of this lane's K fragment kk (slot 0, key half 0)
constexpr int SA_VADDR = 148;  // a148, a149: LDS addresses of the two halves of this lane's V fragment (slot 0, channel tile 0)
constexpr int SA_KOFF = 150;   // a150, a151: LDS-DMA source offsets of this lane's K pieces
constexpr int SA_VOFF = 152;   // a152: lane * 16

template <int R> __device__ __forceinline__ void acc_write(int v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "i"(R)); }
template <int R> __device__ __forceinline__ unsigned acc_read()
{
    unsigned v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R));
    return v;
}
template <int R0> __device__ __forceinline__ void acc_write4(v4i v)
{
    acc_write<R0>(v[0]); acc_write<R0 + 1>(v[1]); acc_write<R0 + 2>(v[2]); acc_write<R0 + 3>(v[3]);
}
// 16 consecutive VGPRs <- 0
template <int R0> __device__ __forceinline__ void vzero16()
{
    asm volatile("v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, 0\n\tv_mov_b32 v%c2, 0\n\tv_mov_b32 v%c3, 0\n\t"
                 "v_mov_b32 v%c4, 0\n\tv_mov_b32 v%c5, 0\n\tv_mov_b32 v%c6, 0\n\tv_mov_b32 v%c7, 0\n\t"
                 "v_mov_b32 v%c8, 0\n\tv_mov_b32 v%c9, 0\n\tv_mov_b32 v%c10, 0\n\tv_mov_b32 v%c11, 0\n\t"
                 "v_mov_b32 v%c12, 0\n\tv_mov_b32 v%c13, 0\n\tv_mov_b32 v%c14, 0\n\tv_mov_b32 v%c15, 0"
                 ::"i"(R0), "i"(R0 + 1), "i"(R0 + 2), "i"(R0 + 3), "i"(R0 + 4), "i"(R0 + 5), "i"(R0 + 6), "i"(R0 + 7),
                   "i"(R0 + 8), "i"(R0 + 9), "i"(R0 + 10), "i"(R0 + 11), "i"(R0 + 12), "i"(R0 + 13), "i"(R0 + 14), "i"(R0 + 15));
}
// ds_read_b128 into a[A0 : A0+3]; completion is the caller's s_waitcnt lgkmcnt
template <int A0, int OFF> __device__ __forceinline__ void lds_read128(unsigned addr)
{
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(A0), "i"(A0 + 3), "i"(OFF));
}
// S^T tile (sb, rb) (+)= K fragment (sb, kk) x Q fragment (rb, kk); kk == 0: C = the bit pattern of 1/(2 pi) (sage_attn.hip, kSInit)
template <int SB, int RB, int KK> __device__ __forceinline__ void mfma_qk()
{
    constexpr int S0 = SV(SB, RB, 0), KR = KA(SB, KK), QR = QA(RB, KK);
    if constexpr (KK == 0)
        asm volatile("v_mfma_i32_32x32x32_i8 v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], 0x3e22f983"
                     ::"i"(S0), "i"(S0 + 15), "i"(KR), "i"(KR + 3), "i"(QR), "i"(QR + 3));
    else
        asm volatile("v_mfma_i32_32x32x32_i8 v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c0:%c1]"
                     ::"i"(S0), "i"(S0 + 15), "i"(KR), "i"(KR + 3), "i"(QR), "i"(QR + 3));
}
// O^T tile (dt, rb) += V fragment dt x P(rb): v_mfma_f32_32x32x64_f8f6f4 is the K = 64 FP8 product sage_attn.hip issues in its
// block-scaled form with unit scales, here without the 8-byte v_mfma_ld_scale prefix and without a scale VGPR
template <int DTI, int RB> __device__ __forceinline__ void mfma_pv()
{
    constexpr int O0 = OV(DTI, RB, 0), VR = VA(DTI), P0 = PV_(RB, 0);
    asm volatile("v_mfma_f32_32x32x64_f8f6f4 v[%c0:%c1], a[%c2:%c3], v[%c4:%c5], v[%c0:%c1]"
                 ::"i"(O0), "i"(O0 + 15), "i"(VR), "i"(VR + 7), "i"(P0), "i"(P0 + 7));
}
// 4 scores -> one P word: bias sub, scale fma, exp2 (g4a); row-sum adds, fp8 pack (g4b).  Two statements so that an MFMA can be
// placed between them (and, on masked tiles, the forced zeros).
template <int R> __device__ __forceinline__ void g4a(float &t0, float &t1, float &t2, float &t3, float ca, float cb, float m)
{
    asm volatile("v_add_f32 %0, 0xbe22f983, v%c7\n\tv_add_f32 %1, 0xbe22f983, v%c8\n\t"
                 "v_add_f32 %2, 0xbe22f983, v%c9\n\tv_add_f32 %3, 0xbe22f983, v%c10\n\t"
                 "v_fma_f32 %0, %0, %4, -%6\n\tv_fma_f32 %1, %1, %4, -%6\n\t"
                 "v_fma_f32 %2, %2, %5, -%6\n\tv_fma_f32 %3, %3, %5, -%6\n\t"
                 "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(ca), "v"(cb), "v"(m), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R> __device__ __forceinline__ void g4b(float t0, float t1, float t2, float t3, float &rs0, float &rs1)
{
    asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                 "v_cvt_pk_fp8_f32 v%c6, %2, %3\n\tv_cvt_pk_fp8_f32 v%c6, %4, %5 op_sel:[0,0,1]"
                 : "+v"(rs0), "+v"(rs1)
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "i"(R));
}
// maxima of one S^T tile's raw scores: m0 over registers i with (i & 2) == 0, m1 over the others (the two per-thread K scale
// groups of the lane, quant_per_thread.py:75-83); FIRST starts the maxima, otherwise it continues them
#ifndef SAGE64_FMAX     // 1: the maxima as v_max3_f32 on the bit patterns (positive normal floats order like their bits)
#define SAGE64_FMAX 0
#endif
#if SAGE64_FMAX
#define SAGE64_MAX3 "v_max3_f32"
#define SAGE64_MAX2 "v_max_f32"
#else
#define SAGE64_MAX3 "v_max3_i32"
#define SAGE64_MAX2 "v_max_i32"
#endif
template <int R, bool FIRST> __device__ __forceinline__ void tile_max(int &m0, int &m1)
{
    if constexpr (FIRST)
        asm volatile(SAGE64_MAX3 " %0, v%c2, v%c3, v%c4\n\t" SAGE64_MAX3 " %1, v%c5, v%c6, v%c7\n\t"
                     SAGE64_MAX3 " %0, %0, v%c8, v%c9\n\t" SAGE64_MAX3 " %1, %1, v%c10, v%c11\n\t"
                     SAGE64_MAX3 " %0, %0, v%c12, v%c13\n\t" SAGE64_MAX3 " %1, %1, v%c14, v%c15\n\t"
                     SAGE64_MAX2 " %0, %0, v%c16\n\t" SAGE64_MAX2 " %1, %1, v%c17"
                     : "=&v"(m0), "=&v"(m1)
                     : "i"(R), "i"(R + 1), "i"(R + 4), "i"(R + 2), "i"(R + 3), "i"(R + 6), "i"(R + 5), "i"(R + 8), "i"(R + 7), "i"(R + 10),
                       "i"(R + 9), "i"(R + 12), "i"(R + 11), "i"(R + 14), "i"(R + 13), "i"(R + 15));
    else
        asm volatile(SAGE64_MAX3 " %0, %0, v%c2, v%c3\n\t" SAGE64_MAX3 " %1, %1, v%c4, v%c5\n\t"
                     SAGE64_MAX3 " %0, %0, v%c6, v%c7\n\t" SAGE64_MAX3 " %1, %1, v%c8, v%c9\n\t"
                     SAGE64_MAX3 " %0, %0, v%c10, v%c11\n\t" SAGE64_MAX3 " %1, %1, v%c12, v%c13\n\t"
                     SAGE64_MAX3 " %0, %0, v%c14, v%c15\n\t" SAGE64_MAX3 " %1, %1, v%c16, v%c17"
                     : "+v"(m0), "+v"(m1)
                     : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3), "i"(R + 4), "i"(R + 5), "i"(R + 6), "i"(R + 7), "i"(R + 8), "i"(R + 9),
                       "i"(R + 10), "i"(R + 11), "i"(R + 12), "i"(R + 13), "i"(R + 14), "i"(R + 15));
}
// masked tiles: score register R + j (j = 0..3, MFMA register index I0 + j of key half SB) keeps its value if its key offset
// 32 SB + ((I0 + j) & 3) + 8 ((I0 + j) >> 2) is <= limv, else becomes the bit pattern of -FLT_MAX (below every biased score
// as an integer; -inf, or the all-zero-row value, once scaled)
template <int R, int SB, int I0> __device__ __forceinline__ void mask4(int limv)
{
    constexpr int K0 = 32 * SB + (I0 & 3) + 8 * (I0 >> 2);
    const int sent = (int)0xFF7FFFFF;           // (a literal next to vcc would be a second constant-bus operand)
    asm volatile("v_cmp_le_i32 vcc, %c5, %0\n\tv_cndmask_b32 v%c1, %9, v%c1, vcc\n\t"
                 "v_cmp_le_i32 vcc, %c6, %0\n\tv_cndmask_b32 v%c2, %9, v%c2, vcc\n\t"
                 "v_cmp_le_i32 vcc, %c7, %0\n\tv_cndmask_b32 v%c3, %9, v%c3, vcc\n\t"
                 "v_cmp_le_i32 vcc, %c8, %0\n\tv_cndmask_b32 v%c4, %9, v%c4, vcc"
                 ::"v"(limv), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3), "i"(K0), "i"(K0 + 1), "i"(K0 + 2), "i"(K0 + 3), "v"(sent) : "vcc");
}
// 16 consecutive VGPRs *= alpha
#ifndef SAGE64_PKMUL    // 1: the rescale as 8 v_pk_mul_f32 per tile instead of 16 v_mul_f32
#define SAGE64_PKMUL 1
#endif
template <int R0> __device__ __forceinline__ void vscale16(float alpha)
{
#if SAGE64_PKMUL
    const v2f a2 = {alpha, alpha};
    asm volatile("v_pk_mul_f32 v[%c1:%c2], %0, v[%c1:%c2]\n\tv_pk_mul_f32 v[%c3:%c4], %0, v[%c3:%c4]\n\t"
                 "v_pk_mul_f32 v[%c5:%c6], %0, v[%c5:%c6]\n\tv_pk_mul_f32 v[%c7:%c8], %0, v[%c7:%c8]\n\t"
                 "v_pk_mul_f32 v[%c9:%c10], %0, v[%c9:%c10]\n\tv_pk_mul_f32 v[%c11:%c12], %0, v[%c11:%c12]\n\t"
                 "v_pk_mul_f32 v[%c13:%c14], %0, v[%c13:%c14]\n\tv_pk_mul_f32 v[%c15:%c16], %0, v[%c15:%c16]"
                 ::"v"(a2), "i"(R0), "i"(R0 + 1), "i"(R0 + 2), "i"(R0 + 3), "i"(R0 + 4), "i"(R0 + 5), "i"(R0 + 6), "i"(R0 + 7),
                   "i"(R0 + 8), "i"(R0 + 9), "i"(R0 + 10), "i"(R0 + 11), "i"(R0 + 12), "i"(R0 + 13), "i"(R0 + 14), "i"(R0 + 15));
    return;
#endif
    asm volatile("v_mul_f32 v%c1, %0, v%c1\n\tv_mul_f32 v%c2, %0, v%c2\n\tv_mul_f32 v%c3, %0, v%c3\n\tv_mul_f32 v%c4, %0, v%c4\n\t"
                 "v_mul_f32 v%c5, %0, v%c5\n\tv_mul_f32 v%c6, %0, v%c6\n\tv_mul_f32 v%c7, %0, v%c7\n\tv_mul_f32 v%c8, %0, v%c8\n\t"
                 "v_mul_f32 v%c9, %0, v%c9\n\tv_mul_f32 v%c10, %0, v%c10\n\tv_mul_f32 v%c11, %0, v%c11\n\tv_mul_f32 v%c12, %0, v%c12\n\t"
                 "v_mul_f32 v%c13, %0, v%c13\n\tv_mul_f32 v%c14, %0, v%c14\n\tv_mul_f32 v%c15, %0, v%c15\n\tv_mul_f32 v%c16, %0, v%c16"
                 ::"v"(alpha), "i"(R0), "i"(R0 + 1), "i"(R0 + 2), "i"(R0 + 3), "i"(R0 + 4), "i"(R0 + 5), "i"(R0 + 6), "i"(R0 + 7),
                   "i"(R0 + 8), "i"(R0 + 9), "i"(R0 + 10), "i"(R0 + 11), "i"(R0 + 12), "i"(R0 + 13), "i"(R0 + 14), "i"(R0 + 15));
}
// 4 consecutive VGPRs -> compiler values
template <int R0> __device__ __forceinline__ void vget4(float (&x)[4])
{
    asm volatile("v_mov_b32 %0, v%c4\n\tv_mov_b32 %1, v%c5\n\tv_mov_b32 %2, v%c6\n\tv_mov_b32 %3, v%c7"
                 : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]) : "i"(R0), "i"(R0 + 1), "i"(R0 + 2), "i"(R0 + 3));
}

#define IC(x) std::integral_constant<int, (x)>{}

template <int D, bool CAUSAL, bool KTHREAD, int QF>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(SAGE64_NCOMP)))
sage_attn64_kernel(const AttnParams p)
{
    using C = Cfg<D>;
    constexpr int KSTEPS = C::KSTEPS, DT = C::DT, PCS = C::PCS;
    constexpr int DMA_PER_TILE = 2 * PCS;                   // LDS-DMA instructions per wave and tile
    static_assert(PCS == 1 || PCS == 2, "LDS-DMA: one or two pieces per wave and image");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the registers named in the text are outside the compiler's budget; this clobber puts the highest of them into the kernel
    // descriptor's register counts
    asm volatile("" ::: "v255", "a159");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;      // query row inside a 32-row block
    const int g = lane >> 5;      // operand half

    // ---- work item: XCD-contiguous runs, longest (causal) blocks of a head first (sage_attn.hip, same map) ----------------
    const int nqblk = (p.Lq + BQ - 1) / BQ;
    int b, h, hk, qblk;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
        const int nwg = gridDim.x;
        const int qq = nwg >> 3, rr = nwg & 7;
        const int wid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        const int bh = wid / nqblk;
        qblk = nqblk - 1 - (wid - bh * nqblk);
        b = bh / p.Hq;
        h = bh - b * p.Hq;
        hk = h / p.group;
    }
    const int Lq = p.Lq, Lk = p.Lk;
    const long q_off = (long)b * p.q_sb + (long)h * p.q_sh;
    const long k_off = (long)b * p.k_sb + (long)hk * p.k_sh;
    const long o_off = (long)b * p.o_sb + (long)h * p.o_sh;
    const int ntk_all = (Lk + BLKK - 1) / BLKK;
    const long v_tile0 = ((long)b * p.Hkv + hk) * ntk_all;
    const float *qs_ptr = p.q_scale + ((long)b * p.Hq + h) * p.nqs;
    const float *ks_ptr = p.k_scale + ((long)b * p.Hkv + hk) * p.nks;
    constexpr int ks_tstride = KTHREAD ? 4 : 1;

    const int row0 = qblk * BQ + wave * 64;                 // first query row of this wave
    int n_iters = ntk_all;                                  // tiles the workgroup walks
    int n_w = ntk_all;                                      // tiles this wave computes on
    if (CAUSAL) {
        const int lim = (qblk * BQ + BQ + BLKK - 1) / BLKK;
        n_iters = lim < n_iters ? lim : n_iters;
        const int limw = (row0 + 64 + BLKK - 1) / BLKK;
        n_w = limw < n_iters ? limw : n_iters;
    }

    // ---- LDS-DMA of one K/V tile (this wave's pieces) --------------------------------------------------------------------
    const unsigned char *kbase = reinterpret_cast<const unsigned char *>(p.k) + k_off;
    const unsigned char *vbase = reinterpret_cast<const unsigned char *>(p.v);
    constexpr int CPR = D / 16;                             // 16-B chunks per K row
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    {
        acc_write<SA_VOFF>(lane * 16);
#pragma unroll
        for (int i = 0; i < PCS; i++) {
            const int e = (wave * PCS + i) * 64 + lane;     // 16-B slot of the tile
            const int row = e / CPR, phys = e % CPR;
            // piece 1 goes out with inst_offset 1024 (which advances the LDS address as well), so its source offset is 1024 less;
            // the offsets are unsigned in the instruction: every offset is biased by +1024 and the base pointer by -1024
            const int o = row * (int)p.k_sl + swz_chunk<D>(row, phys) * 16 + 1024 - i * 1024;
            if (i == 0) acc_write<SA_KOFF>(o); else acc_write<SA_KOFF + 1>(o);
        }
    }
    auto dma_tile = [&](int it, int slot) {
        int tile = it < ntk_all ? it : ntk_all - 1;         // past the end: the last tile again (never consumed)
        const unsigned char *ktp = kbase + (long)tile * BLKK * p.k_sl - 1024;        // (-1024: see the offsets above)
        const unsigned char *vtp = vbase + (v_tile0 + tile) * (long)C::V_BYTES + wave * PCS * 1024;
        const unsigned ldk = lds_base + slot * C::STAGE + wave * PCS * 1024;
        const unsigned ldv = lds_base + slot * C::STAGE + C::K_BYTES + wave * PCS * 1024;
        const unsigned voff16 = acc_read<SA_VOFF>();
        unsigned k0 = acc_read<SA_KOFF>(), k1 = PCS == 2 ? acc_read<SA_KOFF + 1>() : 0u;
        if (tile * BLKK + BLKK > Lk) {                      // ragged tile: rows past Lk read the last valid row (masked scores)
            const int lastrow = Lk - 1 - tile * BLKK;
            const int ln = threadIdx.x & 63;
#pragma unroll
            for (int i = 0; i < PCS; i++) {
                const int e = (wave * PCS + i) * 64 + ln;
                int row = e / CPR;
                const int phys = e % CPR;
                const int sw = swz_chunk<D>(row, phys);
                row = row < lastrow ? row : lastrow;
                const unsigned o = (unsigned)(row * (int)p.k_sl + sw * 16);
                if (i == 0) k0 = o + 1024u; else k1 = o;
            }
        }
        unsigned keep;
        if constexpr (PCS == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %7, %4\n\tglobal_load_lds_dwordx4 %7, %4 offset:1024\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(k0), "v"(k1), "s"(ktp), "s"(vtp), "s"(ldk), "s"(ldv), "v"(voff16) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %6, %3\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(k0), "s"(ktp), "s"(vtp), "s"(ldk), "s"(ldv), "v"(voff16) : "memory");
    };

    // K scales: scalar loads one tile ahead (tracked by lgkmcnt; a VGPR load would make hipcc drain the LDS-DMA, sage_attn.hip).
    // issue: the loads of tile `it`; finish: pins the point where they are waited for (an empty volatile statement that takes
    // the SGPRs -- hipcc would otherwise be free to put its lgkmcnt(0) anywhere in the phases)
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    const cfloat_p ks_c = (cfloat_p)(ks_ptr);
    float ksc[2];
    auto kscales_issue = [&](int it, float (&raw)[4]) {
        int tk = it < ntk_all ? it : ntk_all - 1;
        const long tb = (long)(tk >> p.ks_shift) * ks_tstride;
        if (KTHREAD) { raw[0] = ks_c[tb]; raw[1] = ks_c[tb + 1]; raw[2] = ks_c[tb + 2]; raw[3] = ks_c[tb + 3]; }
        else { raw[0] = ks_c[tb]; raw[1] = raw[2] = raw[3] = 0.0f; }
    };
    auto kscales_finish = [&](float (&raw)[4], float (&dst)[2]) {
        if (KTHREAD) {          // 4 key scales per 64 keys (token % 8 / 2, quant_per_thread.py:75-83); lane half g uses 2g, 2g+1
            asm volatile("" : "+s"(raw[0]), "+s"(raw[1]), "+s"(raw[2]), "+s"(raw[3]));
            dst[0] = g ? raw[2] : raw[0];
            dst[1] = g ? raw[3] : raw[1];
        } else {
            asm volatile("" : "+s"(raw[0]));
            dst[0] = dst[1] = raw[0];
        }
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------
    {
        float raw[4];
        kscales_issue(0, raw);
        kscales_finish(raw, ksc);
    }
#pragma unroll
    for (int i = 0; i < AHEAD; i++) dma_tile(i, i);

    // Q fragments (B operand of S^T = K Q^T) of both row blocks -> a[48:79]; this lane's query-row scales
    float qsc[2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int my_row = row0 + 32 * rb + n;
        const bool ok = my_row < Lq;
        v4i qf[KSTEPS];
        if constexpr (QF == 0) {
            const int8_t *qrow = reinterpret_cast<const int8_t *>(p.q) + q_off + (long)my_row * p.q_sl;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const v4i z = {0, 0, 0, 0};
                qf[ks] = ok ? *reinterpret_cast<const v4i *>(qrow + 32 * ks + 16 * g) : z;
            }
            const int rin = my_row & 127;                    // row inside its 128-row scale block
            int slot;
            if (p.q_gran == QG_PER_BLOCK) slot = 0;
            else if (p.q_gran == QG_PER_WARP32) slot = rin >> 5;
            else if (p.q_gran == QG_PER_WARP16) slot = rin >> 4;
            else if (p.q_gran == QG_PER_THREAD16) slot = (rin >> 4) * 8 + (rin & 7);
            else slot = (rin >> 5) * 8 + (rin & 7);          // per-thread: quant_per_thread.py:27-37
            qsc[rb] = ok ? qs_ptr[(my_row >> 7) * p.qs_per_blk + slot] : 0.0f;
        } else {
            // fused Q quantisation ("per-thread" groups: rows r, r+8, r+16, r+24 of a 32-row block, all channels), sage_attn.hip;
            // one k-step (32 channels) of the row at a time: the prologue shares the compiler's 48 registers
            constexpr int QDT = (QF == 1) ? DT_F16 : DT_BF16;
            const uint16_t *qrow = reinterpret_cast<const uint16_t *>(p.q) + q_off + (long)my_row * p.q_sl;
            auto load16 = [&](int ks, float (&x)[16]) {
                v4u raw[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
                if (ok) {
                    raw[0] = *reinterpret_cast<const v4u *>(qrow + 32 * ks + 16 * g);
                    raw[1] = *reinterpret_cast<const v4u *>(qrow + 32 * ks + 16 * g + 8);
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const unsigned w = raw[j >> 3][(j & 7) >> 1];
                    x[j] = ld16<QDT>((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
                }
            };
            float amax = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                float x[16];
                load16(ks, x);
#pragma unroll
                for (int j = 0; j < 16; j++) amax = fmaxf(amax, fabsf(x[j]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 8));
            amax = fmaxf(amax, __shfl_xor(amax, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const float sc = quant_scale(amax, QS_TRITON_THREAD);
            const float y = quant_recip(sc);
            qsc[rb] = sc;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                float x[16];
                load16(ks, x);                               // second read of the row: an L1 / L2 hit
                int q8[16];
#pragma unroll
                for (int j = 0; j < 16; j++) q8[j] = quant_round_triton_nz(x[j], sc, y);
#pragma unroll
                for (int w = 0; w < 4; w++) qf[ks][w] = (int)pack_int8x4(q8[4 * w], q8[4 * w + 1], q8[4 * w + 2], q8[4 * w + 3]);
            }
        }
        if (rb == 0) {
            acc_write4<QA(0, 0)>(qf[0]);
            if constexpr (KSTEPS > 1) acc_write4<QA(0, 1)>(qf[1]);
            if constexpr (KSTEPS > 2) { acc_write4<QA(0, 2)>(qf[2]); acc_write4<QA(0, 3)>(qf[3]); }
        } else {
            acc_write4<QA(1, 0)>(qf[0]);
            if constexpr (KSTEPS > 1) acc_write4<QA(1, 1)>(qf[1]);
            if constexpr (KSTEPS > 2) { acc_write4<QA(1, 2)>(qf[2]); acc_write4<QA(1, 3)>(qf[3]); }
        }
    }
    // V fragments start as zeros: the first PV(t-1, rb 1) multiplies P = 0 with them and must not meet NaN patterns
    acc_write4<VA(0)>(v4i{0, 0, 0, 0}); acc_write4<VA(0) + 4>(v4i{0, 0, 0, 0});
    acc_write4<VA(1)>(v4i{0, 0, 0, 0}); acc_write4<VA(1) + 4>(v4i{0, 0, 0, 0});
    acc_write4<VA(2)>(v4i{0, 0, 0, 0}); acc_write4<VA(2) + 4>(v4i{0, 0, 0, 0});
    acc_write4<VA(3)>(v4i{0, 0, 0, 0}); acc_write4<VA(3) + 4>(v4i{0, 0, 0, 0});

    // ---- running state -----------------------------------------------------------------------------------------------------
    vzero16<PV_(0, 0)>();                                   // P of both row blocks
    vzero16<OV(0, 0, 0)>(); vzero16<OV(0, 1, 0)>(); vzero16<OV(1, 0, 0)>(); vzero16<OV(1, 1, 0)>();
    vzero16<OV(2, 0, 0)>(); vzero16<OV(2, 1, 0)>(); vzero16<OV(3, 0, 0)>(); vzero16<OV(3, 1, 0)>();
    float m_run[2] = {kNegBig, kNegBig}, l_run[2] = {0.0f, 0.0f};
    const float sm26 = p.sm_scale_log2 * kSUnit;
    constexpr float OFF = kFp8Offset;

    // per-lane LDS read addresses (slot base added per tile)
    //   K fragment (sb, kk): row 32 sb + n, 16-B chunk (2 kk + g) ^ swizzle(n) of the row
    //   V fragment dt: row 32 dt + n of the image, chunks (2g) ^ sw and (2g + 1) ^ sw, sw = (n >> 2) & 3
    acc_write<SA_KADDR>(lds_base + n * D + swz_chunk<D>(n, g) * 16);
    if constexpr (KSTEPS > 1) acc_write<SA_KADDR + 1>(lds_base + n * D + swz_chunk<D>(n, 2 + g) * 16);
    if constexpr (KSTEPS > 2) {
        acc_write<SA_KADDR + 2>(lds_base + n * D + swz_chunk<D>(n, 4 + g) * 16);
        acc_write<SA_KADDR + 3>(lds_base + n * D + swz_chunk<D>(n, 6 + g) * 16);
    }
    acc_write<SA_VADDR>(lds_base + C::K_BYTES + n * 64 + swz_chunk<64>(n, 2 * g) * 16);
    acc_write<SA_VADDR + 1>(lds_base + C::K_BYTES + n * 64 + swz_chunk<64>(n, 2 * g + 1) * 16);

    auto read_k = [&](int slot) {                          // K fragments of the tile in `slot` -> a[80:111]
        const unsigned so = slot * C::STAGE;
        const unsigned a0 = acc_read<SA_KADDR>() + so;
        lds_read128<KA(0, 0), 0>(a0); lds_read128<KA(1, 0), 32 * D>(a0);
        if constexpr (KSTEPS > 1) { const unsigned a1 = acc_read<SA_KADDR + 1>() + so; lds_read128<KA(0, 1), 0>(a1); lds_read128<KA(1, 1), 32 * D>(a1); }
        if constexpr (KSTEPS > 2) {
            const unsigned a2 = acc_read<SA_KADDR + 2>() + so;
            lds_read128<KA(0, 2), 0>(a2); lds_read128<KA(1, 2), 32 * D>(a2);
            const unsigned a3 = acc_read<SA_KADDR + 3>() + so;
            lds_read128<KA(0, 3), 0>(a3); lds_read128<KA(1, 3), 32 * D>(a3);
        }
    };
    auto read_v = [&](int slot) {                          // V fragments of the tile in `slot` -> a[112:143]
        const unsigned so = slot * C::STAGE;
        const unsigned a0 = acc_read<SA_VADDR>() + so, a1 = acc_read<SA_VADDR + 1>() + so;
        lds_read128<VA(0), 0>(a0); lds_read128<VA(0) + 4, 0>(a1);
        lds_read128<VA(1), 2048>(a0); lds_read128<VA(1) + 4, 2048>(a1);
        if constexpr (DT > 2) {
            lds_read128<VA(2), 4096>(a0); lds_read128<VA(2) + 4, 4096>(a1);
            lds_read128<VA(3), 6144>(a0); lds_read128<VA(3) + 4, 6144>(a1);
        }
    };
    // QK^T of one row block: MFMA number j of KSTEPS * 2, k-step-major so that consecutive MFMAs alternate accumulators
    auto qk_step = [&](auto rb_tag, auto j_tag) {
        constexpr int RB = decltype(rb_tag)::value, J = decltype(j_tag)::value;
        constexpr int kk = J >> 1, sb = J & 1;
        if constexpr (kk < KSTEPS && !(SAGE64_ABL & 16)) mfma_qk<sb, RB, kk>();
    };
    auto pv_step = [&](auto rb_tag, auto dt_tag) {
        constexpr int RB = decltype(rb_tag)::value, DTI = decltype(dt_tag)::value;
        if constexpr (DTI < DT && !(SAGE64_ABL & 16)) mfma_pv<DTI, RB>();
    };

    // first tile: wait for it (the Q loads above already drained the DMA queue), K(0) fragments, QK^T(0, rb 0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_k(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
    qk_step(IC(0), IC(0)); qk_step(IC(0), IC(1)); qk_step(IC(0), IC(2)); qk_step(IC(0), IC(3));
    qk_step(IC(0), IC(4)); qk_step(IC(0), IC(5)); qk_step(IC(0), IC(6)); qk_step(IC(0), IC(7));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- one row block's softmax over tile `it` -------------------------------------------------------------------------------
    // FAST: the whole tile is visible to every row of the block.  The caller deals its MFMAs / LDS reads between the statements
    // through mid(j), j = 0 .. 17.  Leaves P in the block's P words, updates m_run / l_run and rescales the block's O tiles when
    // a row maximum moved.
    auto softmax = [&](auto rb_tag, auto fast_tag, const int it, const float (&kscv)[2], auto &&mid) {
        constexpr int RB = decltype(rb_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        float cs[2];                 // (sm * (q_scale * k_scale)) * 2^26, the reference's order of the scale product (sage_attn.hip)
        cs[0] = sm26 * (qsc[RB] * kscv[0]);
        cs[1] = KTHREAD ? sm26 * (qsc[RB] * kscv[1]) : cs[0];
        // masked form: key 64 it + 32 sb + crow(i, g) is visible to this lane's row iff  32 sb + (i & 3) + 8 (i >> 2) <= limv
        int limv = 0;
        if constexpr (!FAST) {
            int lim = Lk - 1;
            if (CAUSAL) { const int r = row0 + 32 * RB + n; lim = r < lim ? r : lim; }
            limv = lim - it * BLKK - 4 * g;
            mask4<SV(0, RB, 0), 0, 0>(limv); mask4<SV(0, RB, 4), 0, 4>(limv); mask4<SV(0, RB, 8), 0, 8>(limv); mask4<SV(0, RB, 12), 0, 12>(limv);
            mask4<SV(1, RB, 0), 1, 0>(limv); mask4<SV(1, RB, 4), 1, 4>(limv); mask4<SV(1, RB, 8), 1, 8>(limv); mask4<SV(1, RB, 12), 1, 12>(limv);
        }
        auto visible = [&](int sb, int i) { return (32 * sb + (i & 3) + 8 * (i >> 2)) <= limv; };
        mid(IC(0));
        int mx0 = 0x3E22F983, mx1 = 0x3E22F983;
        if constexpr (!(SAGE64_ABL & 32)) {
            tile_max<SV(0, RB, 0), true>(mx0, mx1);
            tile_max<SV(1, RB, 0), false>(mx0, mx1);
        }
        float mxc;
        if (KTHREAD) {
            mxc = __builtin_fmaf(__int_as_float(mx0) - __int_as_float(0x3E22F983), cs[0], -OFF);
            mxc = fmaxf(mxc, __builtin_fmaf(__int_as_float(mx1) - __int_as_float(0x3E22F983), cs[1], -OFF));
        } else {
            mxc = __builtin_fmaf(__int_as_float(max(mx0, mx1)) - __int_as_float(0x3E22F983), cs[0], -OFF);
        }
        const float m_new = fmaxf(m_run[RB], pair_max(mxc));
        const float alpha = __builtin_amdgcn_exp2f(m_run[RB] - m_new);
        m_run[RB] = m_new;
        mid(IC(1));
        if (!(SAGE64_ABL & 2) && __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
            vscale16<OV(0, RB, 0)>(alpha);
            if constexpr (DT > 1) vscale16<OV(1, RB, 0)>(alpha);
            if constexpr (DT > 2) { vscale16<OV(2, RB, 0)>(alpha); vscale16<OV(3, RB, 0)>(alpha); }
        }
        float rs0 = 0.0f, rs1 = 0.0f;
        float t0, t1, t2, t3;
#define SAGE_W(w)                                                                                                            \
        if constexpr (!(SAGE64_ABL & 32)) g4a<SV((w) >> 2, RB, 4 * ((w) & 3))>(t0, t1, t2, t3, cs[0], cs[1], m_new);           \
        else { t0 = t1 = t2 = t3 = m_new; }                                                                                   \
        mid(IC(2 + 2 * (w)));                                                                                                 \
        if constexpr (!FAST) {                                                                                                \
            t0 = visible((w) >> 2, 4 * ((w) & 3)) ? t0 : 0.0f;     t1 = visible((w) >> 2, 4 * ((w) & 3) + 1) ? t1 : 0.0f;      \
            t2 = visible((w) >> 2, 4 * ((w) & 3) + 2) ? t2 : 0.0f; t3 = visible((w) >> 2, 4 * ((w) & 3) + 3) ? t3 : 0.0f;      \
        }                                                                                                                     \
        if constexpr (!(SAGE64_ABL & 32)) g4b<PV_(RB, (w))>(t0, t1, t2, t3, rs0, rs1);                                        \
        mid(IC(3 + 2 * (w)));
        SAGE_W(0) SAGE_W(1) SAGE_W(2) SAGE_W(3) SAGE_W(4) SAGE_W(5) SAGE_W(6) SAGE_W(7)
#undef SAGE_W
        l_run[RB] = l_run[RB] * alpha + (rs0 + rs1);
    };

    // ---- the tile loop -----------------------------------------------------------------------------------------------------
    // Ring invariant at the top of tile `it`: tiles it .. it+2 requested (it landed and barrier-synchronised), K(it) in
    // a[80:111], V(it-1) in a[112:143] (zeros for it = 0), S(it, rb 0) complete, P(it-1, rb 1) pending in v[56:63].
#pragma nounroll
    for (int it = 0; it < n_iters; it++) {
        const int slot = it & (NSLOT - 1);
        const int nslot = (it + 1) & (NSLOT - 1);
        // tile it+1 has landed for this wave once at most the youngest group is outstanding; the barrier makes that true for
        // every wave's pieces and says every wave is past its reads of tile it-1, whose slot takes tile it+3
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA_PER_TILE) : "memory");
        if constexpr (!(SAGE64_ABL & 1)) __builtin_amdgcn_s_barrier();
        if constexpr (!(SAGE64_ABL & 4)) dma_tile(it + AHEAD, (it + AHEAD) & (NSLOT - 1));
        if (it < n_w) {
            float ksc_raw[4] = {ksc[0], ksc[1], ksc[0], ksc[1]};
            if constexpr (!(SAGE64_ABL & 64)) kscales_issue(it + 1, ksc_raw);
            const int first0 = row0, first1 = row0 + 32;
            const int last_key = it * BLKK + BLKK - 1;
            const bool fast0 = !(CAUSAL && last_key > first0) && (last_key < Lk);
            const bool fast1 = !(CAUSAL && last_key > first1) && (last_key < Lk);

            // ---- phase A: softmax(it, rb 0) | PV(it-1, rb 1), V(it) reads, QK^T(it, rb 1), K(it+1) reads ----
            auto midA = [&](auto j_tag) {
                constexpr int J = decltype(j_tag)::value;
                if constexpr (DT == 4) {
                    if constexpr (J == 0) pv_step(IC(1), IC(0));
                    else if constexpr (J == 1) pv_step(IC(1), IC(1));
                    else if constexpr (J == 2) pv_step(IC(1), IC(2));
                    else if constexpr (J == 4) pv_step(IC(1), IC(3));
                    else if constexpr (J == 5) { if constexpr (!(SAGE64_ABL & 8)) read_v(slot); }
                    else if constexpr (J == 6) qk_step(IC(1), IC(0));
                    else if constexpr (J == 7) qk_step(IC(1), IC(1));
                    else if constexpr (J == 8) qk_step(IC(1), IC(2));
                    else if constexpr (J == 9) qk_step(IC(1), IC(3));
                    else if constexpr (J == 10) qk_step(IC(1), IC(4));
                    else if constexpr (J == 11) qk_step(IC(1), IC(5));
                    else if constexpr (J == 12) qk_step(IC(1), IC(6));
                    else if constexpr (J == 13) qk_step(IC(1), IC(7));
                    else if constexpr (J == 15) { if constexpr (!(SAGE64_ABL & 8)) read_k(nslot); }
                } else {                                     // D = 64: 2 PV, 4 QK^T
                    if constexpr (J == 0) pv_step(IC(1), IC(0));
                    else if constexpr (J == 2) pv_step(IC(1), IC(1));
                    else if constexpr (J == 4) read_v(slot);
                    else if constexpr (J == 6) qk_step(IC(1), IC(0));
                    else if constexpr (J == 8) qk_step(IC(1), IC(1));
                    else if constexpr (J == 10) qk_step(IC(1), IC(2));
                    else if constexpr (J == 12) qk_step(IC(1), IC(3));
                    else if constexpr (J == 15) read_k(nslot);
                }
            };
            if (fast0) softmax(IC(0), std::true_type{}, it, ksc, midA);
            else softmax(IC(0), std::false_type{}, it, ksc, midA);

            // ---- phase B: softmax(it, rb 1) | PV(it, rb 0), QK^T(it+1, rb 0) ----
            // V(it) fragments have landed once only the K(it+1) reads are outstanding; K(it+1) before the first QK^T
            auto midB = [&](auto j_tag) {
                constexpr int J = decltype(j_tag)::value;
                constexpr int NKR = 2 * KSTEPS;              // ds_reads of read_k
                if constexpr (DT == 4) {
                    if constexpr (J == 0) { asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_nop 1" ::"n"(NKR) : "memory"); pv_step(IC(0), IC(0)); }
                    else if constexpr (J == 1) pv_step(IC(0), IC(1));
                    else if constexpr (J == 2) pv_step(IC(0), IC(2));
                    else if constexpr (J == 4) pv_step(IC(0), IC(3));
                    else if constexpr (J == 6) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); qk_step(IC(0), IC(0)); }
                    else if constexpr (J == 7) qk_step(IC(0), IC(1));
                    else if constexpr (J == 8) qk_step(IC(0), IC(2));
                    else if constexpr (J == 9) qk_step(IC(0), IC(3));
                    else if constexpr (J == 10) qk_step(IC(0), IC(4));
                    else if constexpr (J == 11) qk_step(IC(0), IC(5));
                    else if constexpr (J == 12) qk_step(IC(0), IC(6));
                    else if constexpr (J == 13) qk_step(IC(0), IC(7));
                } else {
                    if constexpr (J == 0) { asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_nop 1" ::"n"(NKR) : "memory"); pv_step(IC(0), IC(0)); }
                    else if constexpr (J == 2) pv_step(IC(0), IC(1));
                    else if constexpr (J == 6) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); qk_step(IC(0), IC(0)); }
                    else if constexpr (J == 8) qk_step(IC(0), IC(1));
                    else if constexpr (J == 10) qk_step(IC(0), IC(2));
                    else if constexpr (J == 12) qk_step(IC(0), IC(3));
                }
            };
            if (fast1) softmax(IC(1), std::true_type{}, it, ksc, midB);
            else softmax(IC(1), std::false_type{}, it, ksc, midB);
            if constexpr (!(SAGE64_ABL & 64)) kscales_finish(ksc_raw, ksc);
        }
    }
    // drain: PV(n_w - 1, rb 1) (its V fragments are still in a[112:143]); every LDS-DMA must have landed before the epilogue
    // reuses the ring as the output staging area
    asm volatile("s_nop 1" ::: "memory");
    pv_step(IC(1), IC(0)); pv_step(IC(1), IC(1)); pv_step(IC(1), IC(2)); pv_step(IC(1), IC(3));
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: normalise, x v_scale (+ v_mean), cast, transpose through the wave's own LDS region, store rows ----------
    unsigned char *obuf = smem + wave * (64 * D * 2);
    const float *vsc = p.v_scale + ((long)b * p.Hkv + hk) * D;
    const float *vmn = (p.v_mean != nullptr) ? p.v_mean + ((long)b * p.Hkv + hk) * D : nullptr;
    float inv[2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int my_row = row0 + 32 * rb + n;
        const float l_tot = pair_sum(l_run[rb]);
        inv[rb] = l_tot > 0.0f ? __builtin_amdgcn_rcpf(l_tot) : 0.0f;
        if (p.lse != nullptr && g == 0 && my_row < Lq)
            p.lse[((long)b * p.Hq + h) * (long)p.Lq + my_row] = __builtin_amdgcn_logf(l_tot) + m_run[rb];   // v_log_f32 is log2
    }
    // one 32-channel tile at a time (the per-channel factors of a tile are 2 x 16 registers of the compiler's 48)
    auto epi_tile = [&](auto dt_tag) {
        constexpr int dt = decltype(dt_tag)::value;
        if constexpr (dt < DT) {
            v4f sc4[4], mn4[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) sc4[r4] = *reinterpret_cast<const v4f *>(vsc + dt * 32 + 8 * r4 + 4 * g);
            if (vmn != nullptr) {
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) mn4[r4] = *reinterpret_cast<const v4f *>(vmn + dt * 32 + 8 * r4 + 4 * g);
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) { const v4f z = {0.0f, 0.0f, 0.0f, 0.0f}; mn4[r4] = z; }
            }
            auto quarter = [&](auto rb_tag, auto r4_tag) {
                constexpr int rb = decltype(rb_tag)::value, r4 = decltype(r4_tag)::value;
                float x[4];
                vget4<OV(dt, rb, 4 * r4)>(x);
#pragma unroll
                for (int j = 0; j < 4; j++) x[j] = x[j] * inv[rb] * sc4[r4][j] + mn4[r4][j];
                v2u pk;
                if (p.out_dtype == DT_F16) {
                    pk[0] = (unsigned)f32_to_f16_rne(x[0]) | ((unsigned)f32_to_f16_rne(x[1]) << 16);
                    pk[1] = (unsigned)f32_to_f16_rne(x[2]) | ((unsigned)f32_to_f16_rne(x[3]) << 16);
                } else {
                    pk[0] = (unsigned)f32_to_bf16_rne(x[0]) | ((unsigned)f32_to_bf16_rne(x[1]) << 16);
                    pk[1] = (unsigned)f32_to_bf16_rne(x[2]) | ((unsigned)f32_to_bf16_rne(x[3]) << 16);
                }
                const int d0 = dt * 32 + 8 * r4 + 4 * g;
                const int q8 = d0 >> 2;
                const int Q = (q8 >> 1) ^ (n & 7);
                *reinterpret_cast<v2u *>(obuf + (32 * rb + n) * (D * 2) + Q * 16 + (q8 & 1) * 8) = pk;
            };
            quarter(IC(0), IC(0)); quarter(IC(0), IC(1)); quarter(IC(0), IC(2)); quarter(IC(0), IC(3));
            quarter(IC(1), IC(0)); quarter(IC(1), IC(1)); quarter(IC(1), IC(2)); quarter(IC(1), IC(3));
        }
    };
    epi_tile(IC(0)); epi_tile(IC(1)); epi_tile(IC(2)); epi_tile(IC(3));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        constexpr int LPR = D * 2 / 16;          // lanes per row (16 B each)
        constexpr int RPP = 64 / LPR;            // rows per pass
        unsigned char *obase = reinterpret_cast<unsigned char *>(p.o) + 2 * o_off;
#pragma unroll
        for (int pass = 0; pass < 64 / RPP; pass++) {
            const int r = pass * RPP + lane / LPR, Q = lane % LPR;
            const v4u val = *reinterpret_cast<const v4u *>(obuf + r * (D * 2) + (Q ^ (r & 7)) * 16);
            const int grow = row0 + r;
            if (grow < Lq) *reinterpret_cast<v4u *>(obase + 2 * ((long)grow * p.o_sl) + Q * 16) = val;
        }
    }
}
#undef IC

template <int D, bool CAUSAL, bool KTHREAD, int QF>
static hipError_t launch_one(const AttnParams &p, hipStream_t stream)
{
    using C = Cfg<D>;
    auto kern = sage_attn64_kernel<D, CAUSAL, KTHREAD, QF>;
    const int nwork = p.B * p.Hq * ((p.Lq + BQ - 1) / BQ);
    if (nwork <= 0) return hipSuccess;
    if (C::LDS >= 65536) {      // (64 KiB exactly: opt in as well, the default limit is not documented as inclusive)
        static thread_local unsigned long long done_mask = 0;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done_mask & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
            if (e != hipSuccess) return e;
            done_mask |= bit;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(256), C::LDS, stream, p);
    return hipGetLastError();
}

}  // namespace a64

// FP8 PV, dense, unmasked, D = 128: the one-wave-per-SIMD kernel.  qf: 0 = INT8 q + q_scale, 1 / 2 = fp16 / bf16 q quantised in
// the prologue (per-thread groups).
hipError_t launch_attn64(const AttnParams &p, int head_dim, bool causal, bool kthread, int qf, hipStream_t stream)
{
    if (head_dim != 128 || p.cu_q != nullptr || p.kv_split > 1 || qf < 0 || qf > 2 || (qf != 0 && !kthread)) return hipErrorInvalidValue;
#define SAGE_C64(C_, K_, Q_) if (causal == C_ && kthread == K_ && qf == Q_) return a64::launch_one<128, C_, K_, Q_>(p, stream);
    SAGE_C64(false, true, 0) SAGE_C64(true, true, 0) SAGE_C64(false, false, 0) SAGE_C64(true, false, 0)
    SAGE_C64(false, true, 1) SAGE_C64(true, true, 1) SAGE_C64(false, true, 2) SAGE_C64(true, true, 2)
#undef SAGE_C64
    return hipErrorInvalidValue;
}

}  // namespace sage
