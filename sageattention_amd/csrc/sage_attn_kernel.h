// sage_attn_kernel.h -- fused INT8-QK^T / online-softmax / FP8-or-FP16-PV attention for gfx950: the kernel family and its launcher.
// Included by the instantiation units sage_attn_d{128,64}_{f8,f8f,f16}.hip (one per head size, PV format and FP8 score form, so that they
// compile in parallel); sage_attn.hip holds the host-side dispatch.
//
// Replaces (behaviourally, not textually) the reference kernels
//   csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:46-704   (INT8 QK, FP8 PV, two-level accumulation)
//   csrc/qattn/qk_int_sv_f8_cuda_sm90.cu:127-567   (same, 128-key tiles, RO += RO_temp per tile)
//   csrc/qattn/qk_int_sv_f16_cuda_sm80.cu:46-671   (INT8 QK, FP16 PV)
//   sageattention/triton/attn_qk_int8_per_block*.py, attn_qk_int8_block_varlen.py (+causal)
// with one CDNA4 kernel family.  Design:
//
//  * workgroup = 4 waves = 128 query rows of one (batch, q-head); wave w owns rows 32w..32w+31.
//  * swapped product S^T = K Q^T on v_mfma_i32_32x32x32_i8: A = K tile rows from LDS, B = Q
//    fragments kept in VGPRs for the whole kernel.  In the 32x32 C layout a lane then holds 16
//    keys of ONE query row (col = lane&31), so row max / row sum are in-lane chains plus one
//    v_permlane32_swap with the lane^32 partner.
//  * P is converted in registers (v_cvt_pk_fp8_f32 / cvt f16) and is already the B operand of
//    O^T = V^T P^T: the V pre-pass stores V^T tiles in the matching "position" order
//    (sage_common.h), so P never goes through LDS.  FP8 PV runs on the K = 64 instruction
//    v_mfma_f32_32x32x64_f8f6f4 (the plain form: no block scales), twice the rate of the 32x32x16 fp8 MFMA.
//  * one loop iteration covers one 64-key image; scales and masks are per 64-key block.  Two-level
//    accumulation: whole unmasked tiles add their P.V product to the FP32 running output through the
//    MFMA's FP32 C operand (O = O*alpha + sum p v); general tiles start from a zero accumulator and fold
//    it in with one FMA per element (O = O*alpha + T).
//  * K/V tiles live in a 3-slot LDS ring and arrive by LDS-DMA (global_load_lds_dwordx4) two tiles ahead; the
//    K image is XOR-swizzled through the per-lane SOURCE address, the V image is pre-swizzled
//    by the pre-pass: every MFMA operand read is a conflict-free ds_read_b128.
//  * whole unmasked tiles run software-pipelined loops whose instruction order is pinned in asm (six bodies
//    per trip: ring slot and score-register set are compile-time constants in each).  A work item's last tiles
//    -- the two diagonal tiles of a causal block, the last whole tiles and the ragged tile of a non-causal call --
//    run the same body in three more KINDs behind the loop (nothing more requested, scores behind the diagonal or
//    past Lk masked in front of the row maximum; DIAG_PIPE / TAIL_PIPE below say for which instantiations); what
//    is left -- attn_mask variants, causal blocks ending in a partial tile, ragged FP16-PV tails, D = 64 FP16 PV,
//    sequences under two tiles -- runs the general, phased iteration (tile_iter).
//  * output tile is transposed through (now free) LDS and stored as whole rows, 16 B per lane.
#pragma once
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_attn_parts.h"
#include "sage_quant_math.h"
#include "sage_work_order.h"
#include <atomic>
#include <climits>
#include <cstdlib>
#include <type_traits>

// ---- build-time switches --------------------------------------------------------------------------
// Rounds 1-5 carried an A/B ladder of ~20 switches here (SAGE_GLDS, SAGE_MXPV, SAGE_KPRELOAD, SAGE_STEADY, SAGE_STAGES, SAGE_NH_F8, SAGE_MAGIC,
// SAGE_PIPE, SAGE_PIPE16, SAGE_PLAIN_PV, SAGE_GRP4, SAGE_RSUM_MFMA, SAGE_FOLDBIAS, SAGE_DIRECT, SAGE_PERS_QF, SAGE_PERS_CAUSAL); every one is
// folded to the value that shipped and its alternative body deleted (round 6).  The last tree that still builds them:
// `git show b0d43f7:sageattention_amd/csrc/sage_attn_kernel.h`; what each measured is in DESIGN.md 3.1 / 3.7 / 3.8.  What is fixed now: K / V tiles
// by LDS-DMA into a 3-slot ring; 64-key iterations; the INT32 QK^T accumulators start from the bit pattern of the inline constant
// 1 / (2 pi) = 0x3E22F983 (free as the MFMA's C operand), so read as a float they are 1 / (2 pi) + s * 2^-26 exactly and one exact v_add_f32
// replaces v_cvt_f32_i32; software-pipelined steady-state loops for FP8 and FP16 PV with their instruction order pinned in asm; FP8 PV on
// v_mfma_f32_32x32x64_f8f6f4; two-level requests accumulate P.V through the MFMA's FP32 C operand and rescale O only where a row maximum of
// the wave moved; the ticket loop in the non-causal kernels and in the packed route's causal ones.
#ifndef SAGE_ABL             // timing ablations of the FP8 pipelined loop (WRONG results; tools/build_variants.sh): 1 no O rescale, 2 no s_nop in
#define SAGE_ABL 0           // front of the loop's MFMAs (since they were dropped: 2 = WITH them), 4 no per-tile barrier, 8 no row-maximum chain, 16 no exponentials (v_mov instead), 32 no wait for the next tile's LDS-DMA at the top of a body
#endif
#ifndef SAGE_DIAG_PIPE       // causal FP8 D = 128: a work item's last two tiles through the pipelined body (1) or as general iterations (0: A/B)
#define SAGE_DIAG_PIPE 1
#endif
#ifndef SAGE_KARG_PREFETCH   // one scalar load per line of the parameter block at kernel entry (1) or not (0: A/B)
#define SAGE_KARG_PREFETCH 1
#endif
#ifndef SAGE_TAIL_PIPE       // non-causal FP8: the last two whole tiles (+ a ragged one behind them) through the pipelined body (1) or as general iterations (0: A/B)
#define SAGE_TAIL_PIPE 1
#endif
#ifndef SAGE_KSEL            // pipelined loops, per-thread k scale groups: the lane halves' scale products under EXEC (1) or by select (0: A/B)
#define SAGE_KSEL 1
#endif
#ifndef SAGE_ATTN_TRACE      // tools/attn_trace.py: wave 0 of every workgroup records 100 MHz time stamps of its phases
#define SAGE_ATTN_TRACE 0    // (entry, geometry known, Q ready, first tile landed, key loop done, epilogue barrier, stores issued, stores acknowledged)
#endif

// __launch_bounds__ waves / SIMD the register allocator must allow: the software-pipelined loops carry two score tiles, which at D = 128 is
// 2 waves (248 VGPRs); D = 64 fits 3 (167); the attn_mask variants 2.
#define SAGE_MIN_WAVES(D, MASK) ((MASK) != 0 ? 2 : ((D) == 64 ? 3 : 2))

// asm text of the pipelined loops: two scores d0 / d1 from the bit patterns s0 / s1 of the QK^T accumulators, d = score * c - m (operands as
// asm placeholders).  EXACT: the bias of the bit pattern is subtracted first (exact), then the FMA; FOLD (the FP8 opt-in variant): one FMA per
// score, m already carries the bias
#define SAGE_SCALE2_FOLD(d0, d1, s0, s1, c0, c1, m) "v_fma_f32 " d0 ", " s0 ", " c0 ", -" m "\n\tv_fma_f32 " d1 ", " s1 ", " c1 ", -" m "\n\t"
#define SAGE_SCALE2_EXACT(d0, d1, s0, s1, c0, c1, m) "v_add_f32 " d0 ", 0xbe22f983, " s0 "\n\tv_add_f32 " d1 ", 0xbe22f983, " s1 "\n\t" \
                                                      "v_fma_f32 " d0 ", " d0 ", " c0 ", -" m "\n\tv_fma_f32 " d1 ", " d1 ", " c1 ", -" m "\n\t"

// The pipelined loops' rename of the score tiles, set B -> set A (sA, sB: v16i[2] in scope), behind a body whose last instructions are the
// asm-issued MFMAs that write sB.  Plain copies (sA = sB) are moves the compiler is free to place right behind that asm -- inside the MFMAs'
// latency, which it does not see (round 5's wrong rows; an in-out nop statement in front of the copies does not help: the allocator may
// satisfy its tie by copying first).  So the copy is issued from asm as well, on the matrix pipe: D = 0 * 0 + C moves sixteen registers per
// instruction, exactly (INT32), behind the wait states an MFMA reading another MFMA's result as SrcC needs, and followed by those a VALU
// reader of its own result needs (tools/mfma_hazard_lint.py checks both).
#define SAGE_RENAME_S() do { const v4i z4_ = {0, 0, 0, 0};                                                                   \
        asm volatile("s_nop 15\n\ts_nop 7\n\tv_mfma_i32_32x32x32_i8 %0, %2, %2, %3\n\tv_mfma_i32_32x32x32_i8 %1, %2, %2, %4\n\t"     \
                     "s_nop 15\n\ts_nop 7" : "=&v"(sA[0]), "=&v"(sA[1]) : "v"(z4_), "v"(sB[0]), "v"(sB[1])); } while (0)

namespace sage {

// hwreg(HW_REG_MODE, 23, 1): the FP16_OVFL bit of the MODE register (id 1 | offset 23 << 6 | (width 1 - 1) << 11)
constexpr int kHwregModeFp16Ovfl = 1 | (23 << 6) | (0 << 11);


template <int D, bool PV_FP8, int NH> struct TileCfg {
    static constexpr int KT = BLKK * NH;                        // keys per iteration
    static constexpr int K_TILE_BYTES = KT * D;                 // int8
    static constexpr int V_ROW_BYTES = PV_FP8 ? 64 : 128;       // one 64-key image row
    static constexpr int V_IMG_BYTES = D * V_ROW_BYTES;
    static constexpr int STAGE_BYTES = K_TILE_BYTES + NH * V_IMG_BYTES;
    static constexpr int O_BYTES = BLKQ * D * 2;
    static constexpr int NSTAGE = 3;                            // LDS ring depth: two tiles in flight, counted vmcnt + raw s_barrier
    static constexpr int LDS_BYTES = (NSTAGE * STAGE_BYTES > O_BYTES) ? NSTAGE * STAGE_BYTES : O_BYTES;
    static constexpr int KSTEPS = D / 32;                       // i8 MFMA k-steps over head dim
    static constexpr int DT = D / 32;                           // 32-wide output d tiles
};

// c/d register r of a 32x32 MFMA tile -> row index inside the tile (lane half g = lane>>5)
__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// raw QK^T accumulator -> float score (in units of 2^-kSUnitLog2).
// 0x3E22F983 lies mid-binade ([0.125, 0.25), ulp 2^-26, mantissa field 2292099): for |s| <= 128 * 128 * 128 = 2097152 the sum
// stays inside the binade, so bits + s is the float 1/(2 pi) + s * 2^-26, the subtraction below is exact (Sterbenz) and
// fma(s * 2^-26, c * 2^26, -m) rounds the same real number as fma((float)s, c, -m): bit-identical to the conversion.
constexpr int kSUnitLog2 = 26;              // 2^26 is folded into the score scale -- by v_ldexp_f32 (exact, as a multiplication by 2^26 is, and the
                                            // exponent is an inline operand: the constant 2^26 in a VGPR was one the D = 64 instantiations spilled)
__device__ __forceinline__ float sfl(int x) { return __int_as_float(x) - __int_as_float(0x3E22F983); }
// first MFMA of a QK^T accumulation chain: C = kSInit as an inline constant (hipcc materialises an integer splat of
// 0x3E22F983 in 16 VGPRs instead; the assembler encodes it as inline operand 248).  The builtin MFMAs that follow take the
// result whole as their C operand (accumulate chain: no wait states, cdna_hip_programming.md 5.7 item 2).
__device__ __forceinline__ v16i mfma_i8_first(v4i a, v4i b)
{
    v16i d;
    asm("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 0x3e22f983" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}

#if SAGE_ATTN_TRACE
// (the stamps go to AttnParams::trace, a caller-owned buffer of 16 words per logical workgroup: 8 stamps, -, HW_ID, XCC_ID, blockIdx.x, query block --
//  SageLaunchAttr::trace / trace_wgs, read by trace builds only)
#define SAGE_TSTAMP(i) do { if (wave == 0) { unsigned long long t_; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ttrace[i] = (unsigned)t_; } } while (0)
#else
#define SAGE_TSTAMP(i) do { } while (0)
#endif

// QF: 0 = q is INT8 with scales in q_scale; 1 / 2 = q is fp16 / bf16 and is quantised in the prologue, per-thread groups;
// 3 / 4 = fp16 / bf16 quantised in the prologue PER BLOCK of 128 rows after the multiplication by p.q_premul (quant_per_block.py:21-46
// with sm_scale folded in: the Q half of the reference's Triton-named API and of sageattn_varlen),
// ("per-thread" groups, quant_per_thread.py:21-52), so the INT8 copy of Q and its scales never touch HBM.
// SFOLD (FP8 PV only): false = the exact score form, exp2(fma(s, c, -m)) with the bias of the score's bit pattern subtracted first -- the
// reference's formula (attn_utils.cuh:445-449), the default of every entry point; true = the opt-in variant SAGE_ATTR_FP8_FOLDED_SCORES, the bias
// folded into the scale FMA in every tile of the launch (exp2(fma(bits, c', -(m + bias c'))): one VALU instruction less per score, m + bias c'
// rounded once per (row, tile, k scale); the oracle's score_mode 1 mirrors it).  FP16-PV instantiations have one form, the exact one, and pass true.
// CPERS: the persistent ticket loop compiled into a CAUSAL instantiation (the packed route's launches over the work list; non-causal unmasked
// instantiations always carry it).
// VROWS (FP16 PV, dense): p.v is the caller's fp16 V tensor itself, rows of D halves with element strides p.v_sb / v_sh / v_sl -- what the
// reference's kernels take (value fp16, last dimension contiguous: qk_int_sv_f16_cuda_sm80.cu:693-704) -- instead of the pre-transposed tile
// image of sage_prep_v_f16 / sage_prepass_kv.  A 64-token tile lands in LDS as rows (LDS-DMA through per-lane source addresses, like K) and
// the PV MFMA's A operand -- V^T: lane = channel, 8 tokens -- comes out of two transposing reads, ds_read_b64_tr_b16, which hand lane i of a
// 16-lane group element (i & 3) of the four 8-byte chunks that lanes (i >> 2), 4 + (i >> 2), 8 + .., 12 + .. address
// (tools/microbench/ubench9_tr_b16.hip, profiles/r6_run_c_ubench9_tr_b16.txt).  For fp16 inputs the V half of the pre-pass -- 2/3 of its
// bytes on an FP16-PV call -- disappears; the outputs are bit-identical to the image route's (same operands, same MFMAs).
template <int D, bool PV_FP8, bool CAUSAL, bool KTHREAD, bool TWO_LEVEL, int NH, int MASK = 0, int QF = 0, bool SFOLD = true, bool CPERS = false,
          bool VROWS = false>
__global__ void __launch_bounds__(256, SAGE_MIN_WAVES(D, MASK))
sage_attn_kernel(const AttnParams p_arg)
{
    // The parameter block is read through the kernarg segment pointer, and inside the persistent loop through a copy of that pointer the
    // compiler cannot see through (an empty asm): otherwise every scalar load of a parameter is hoisted out of the loop and stays live in
    // SGPRs across it (128 SGPRs and 16-400 VGPRs spilled in every instantiation).  AttnParams is the kernel's only explicit argument.
    typedef const __attribute__((address_space(4))) AttnParams *kparams_t;
    const kparams_t kp0 = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) AttnParams &p = *kp0;
    (void)p_arg;
#if SAGE_KARG_PREFETCH
    // The parameter block spans seven 64-byte lines and the prologue reads it in ten dependent rounds of scalar loads (branches in between): in a
    // launch's first round of workgroups every first touch of a line is a miss of the scalar cache, three or four of them in series.  One load per
    // line here, waited for together: one miss time instead, the rounds below hit.  (Values unused; each load has its own destination.)
    static_assert(sizeof(AttnParams) > 0x180, "prefetch offsets lie inside the parameter block");
    {
        unsigned t0, t1, t2, t3, t4, t5, t6;
        asm volatile("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %7, 0x40\n\ts_load_dword %2, %7, 0x80\n\ts_load_dword %3, %7, 0xc0\n\t"
                     "s_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\ts_load_dword %6, %7, 0x180\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6) : "s"(kp0) : "memory");
    }
#endif
    using C = TileCfg<D, PV_FP8, NH>;
    constexpr int KT = C::KT;
    constexpr int NS = 2 * NH;                       // 32-key S^T sub-tiles per iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // (wave index in an SGPR, lane index from v_mbcnt wherever it is needed: nothing derived from threadIdx.x has to stay in a VGPR across
    //  the persistent loop below)
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#if SAGE_ATTN_TRACE
    __shared__ unsigned ttrace[16];
#endif
    // ---- persistent launch (p.sched != null; non-causal, unmasked instantiations only): gridDim.x workgroups -- as many as the device holds at
    //      once -- work through the logical grid of p.nwg workgroup indices.  The indices are dealt into 32 queues: index i belongs to XCD i & 7
    //      (the work order's L2 locality, sage_work_order.h) and there to sub-queue (i >> 3) & 3, i.e. i = 32 k + 8 s + x.  A workgroup starts with
    //      its own blockIdx.x (the hardware deals blockIdx.x to XCD blockIdx.x & 7) and then takes tickets k from the counter of its queue -- the
    //      ticket is requested behind the last tile of the item in hand and read after its output rows are on their way.  When that queue is empty
    //      it looks at all 32 counters once and takes a ticket from the fullest queue, its own XCD's first (the XCDs of a device run a few per cent
    //      apart: profiles/r4_run_p_attention_phase_trace.txt).  One counter per 128-byte line, zeroed by the caller: agent-scope atomics on one
    //      address serialise at ~200 ns each, and the 64 workgroups of an XCD finish equal items together.
    //      Causal launches keep the hardware's dispatch: their work order pairs a long and a short block on a CU through the order in which
    //      freed slots are refilled, and tickets lose that (measured: +2.6 % at C3, +7 % at C2).
    // (round 5: the packed / varlen route's CAUSAL launches over the device-built work list take the route too -- +2.2 ... 2.9 % at C4 -- through
    //  instantiations of their own (CPERS); dense causal launches lose 0.1 ... 7.5 % with tickets and the loop's mere presence costs the dense
    //  Triton-API causal kernel 1.3 %, so their instantiations stay without it: profiles/r5_pers_causal_probe.txt, r5_run_c_qf_pers_ab.txt)
    constexpr bool PERS_OK = (!CAUSAL || CPERS) && MASK == 0;
    const bool pers = PERS_OK && p.sched != nullptr;
    __shared__ int s_ticket[2];                 // (two slots, alternating: a wave may still be reading the previous ticket when wave 0 posts the next)
    int tpar = 0;
    int bid = blockIdx.x;
    unsigned next_k_v = 0;                       // (lane 0 of wave 0) the ticket requested ahead
    bool have_next = false, own_empty = false;   // wave-uniform
    // this workgroup's queue: 4 * XCD (HW_REG_XCC_ID[3:0]) + sub-queue.  Everything the ticket code needs besides the three words above is
    // re-derived where it is used (a register read, a scalar load), so that nothing of it is live across the key loop: two SGPRs more
    // spilled to a VGPR cost the D = 64 per-thread instantiation its last register under the three-waves limit.
    // (parameters through a pointer the compiler cannot see through, as in the item loop below: else their scalar loads are hoisted and stay live)
    auto kpl = [&]() -> kparams_t { kparams_t k = kp0; asm volatile("" : "+s"(k)); return k; };
    auto my_queue = [&]() -> int { return 4 * (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) + (int)((blockIdx.x >> 3) & 3u); };
    // the logical grid
    auto logical_grid = [&]() -> int {
        const __attribute__((address_space(4))) AttnParams &pl = *kpl();
        int n = pl.nwg;
        if (pl.cu_q != nullptr && pl.work_items != nullptr) {
            typedef const __attribute__((address_space(4))) int *cint_p;
            const cint_p hdr = (cint_p)pl.work_hdr;
            n = 8 * (hdr[3] * ((hdr[0] + 7) >> 3) + (pl.Hq >> 3) * hdr[0]);
        }
        return n;
    };
    // wave 0: the next logical workgroup index, or -1 when every queue is empty
    auto resolve_ticket = [&]() -> int {
        const int nwg_l = logical_grid();
        const int sched_first = gridDim.x >> 5;      // tickets of every queue that the first round (blockIdx.x) covers
        const int my_q = my_queue(), my_xcd = my_q >> 2;
        unsigned *const sched = kpl()->sched;
        if (have_next) {
            have_next = false;
            const int i = 32 * ((int)__builtin_amdgcn_readfirstlane(next_k_v) + sched_first) + 8 * (my_q & 3) + my_xcd;
            if (i < nwg_l) return i;
            own_empty = true;
        }
        int lane_o = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_o));         // (or the per-lane queue addresses below are computed once, outside the item loop, and stay live)
        for (int attempt = 0; attempt < 4; attempt++) {
            // lane q < 32 looks at queue q; the fullest queue wins, queues of the own XCD before the others
            const int q = lane_o & 31, x = q >> 2, s8x = 8 * (q & 3) + x;
            const unsigned c = lane_o < 32 ? __hip_atomic_load(sched + 32 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const int cnt = nwg_l > s8x ? (nwg_l - s8x + 31) >> 5 : 0;
            int left = cnt - sched_first - (int)c;
            left = left < (1 << 24) ? left : (1 << 24) - 1;
            unsigned key = (lane_o < 32 && left > 0) ? ((x == my_xcd ? 1u : 0u) << 30) | ((unsigned)left << 5) | (unsigned)q : 0u;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { const unsigned o = __shfl_xor(key, m); key = o > key ? o : key; }
            key = __builtin_amdgcn_readfirstlane(key);
            if (key == 0u) return -1;
            const int bq = (int)(key & 31u);
            unsigned kv = 0;
            if (lane_o == 0) kv = __hip_atomic_fetch_add(sched + 32 * bq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int i = 32 * ((int)__builtin_amdgcn_readfirstlane(kv) + sched_first) + 8 * (bq & 3) + (bq >> 2);
            if (i < nwg_l) return i;
        }
        return -1;
    };
    while (bid >= 0) {
    next_k_v = 0;                                // (defined at the top of every pass: not carried round the loop in a VGPR)
    do {
    kparams_t kp = kp0;
    if constexpr (PERS_OK) asm volatile("" : "+s"(kp));
    const __attribute__((address_space(4))) AttnParams &p = *kp;
    // (the same for everything derived from the thread index: hoisted out of the loop, the prologue's and the epilogue's per-lane
    //  offsets would stay live through the key loop -- 13-32 VGPRs spilled in every instantiation)
    int lane_v = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (PERS_OK) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_v));     // (per item: the builtin's value is hoisted out of the loop and lives through it)
    const int lane = lane_v;
    const int wave = wave_s;
    [[maybe_unused]] const int tid = wave * 64 + lane;
    int n = lane & 31;            // query row inside the wave's 32-row tile
    int g = lane >> 5;            // k-group (operand half)
    SAGE_TSTAMP(0);
    // Inside the key loop the conversions of P saturate to the largest finite value instead of overflowing (MODE.FP16_OVFL = 1; back to 0 in front
    // of the epilogue, whose output conversion overflows to inf as the reference's does).  FP8 PV: the reference converts P with
    // cvt.rn.satfinite.e4m3x2.f32 (numeric_conversion.cuh:46-61) and v_cvt_pk_fp8_f32 without the mode bit returns NaN above 464.  With the exact
    // score form P exceeds 448 = 2^8.807 only by the rounding of m (scores of millions); the folded form's m + bias c' is rounded at a magnitude
    // of bias c', so from c = sm_scale log2(e) q_scale k_scale ~ 0.1 (|q|, |k| ~ 100) on a row's largest P can pass 464 -- NaN rows without this.
    // FP16 PV: the folded bias of the pipelined loops likewise, at c ~ 50, against fp16's 65504.
    __builtin_amdgcn_s_setreg(kHwregModeFp16Ovfl, 1);

    // ---- work item: XCD-aware, heavy-first --------------------------------------------------
    const int nqblk = p.nqblk;
    int b, h, hk, qblk;
    if (p.cu_q != nullptr && p.work_items != nullptr) {
        // varlen with the device-built work list (sage_varlen_plan): the query blocks of all sequences are one item list per query head,
        // heaviest first, and the launch is a dense launch over Hq heads of `nitems` items each (sage_work_order.h) -- whole GQA groups
        // stay on one XCD, every workgroup has an item (the grid is sized by a host-known bound of nitems; the few past it exit here)
        typedef const __attribute__((address_space(4))) int *cint_p;          // wave-uniform: scalar loads
        const cint_p hdr = (cint_p)p.work_hdr;
        const int nitems = hdr[0];
        const WorkOrder wo = {hdr[1], hdr[2], hdr[3]};
        const int nwg = 8 * (wo.left * ((nitems + 7) >> 3) + (p.Hq >> 3) * nitems);
        int qrank;
        if (bid >= nwg || !work_item(wo, bid, nwg, p.Hq, nitems, h, qrank)) break;
        const cint_p items = (cint_p)p.work_items;
        b = items[2 * qrank];
        qblk = items[2 * qrank + 1];
        hk = h / p.group;
    } else if (p.cu_q != nullptr) {
        // varlen without a work list (more sequences than sage_varlen_plan takes): sequences differ in length, so a contiguous run
        // per XCD would hand one XCD the longest sequence
        // (measured 3.5x slower on lengths 256..16384).  XCDs take (sequence, kv-head) units round-robin instead;
        // inside a unit the `group` query heads that share the K/V stream run heavy-first, interleaved.
        const int xcd = bid & 7, idx = bid >> 3;
        const int per_unit = nqblk * p.group;
        const int j = idx / per_unit, within = idx - j * per_unit;
        const int u = j * 8 + xcd;
        if (u >= p.B * p.Hkv) break;
        const int r = within / p.group, hg = within - r * p.group;
        qblk = nqblk - 1 - r;
        const int bs = u / p.Hkv;
        hk = u - bs * p.Hkv;
        b = p.seq_order != nullptr ? p.seq_order[bs] : bs;      // caller's processing order (longest first)
        h = hk * p.group + hg;
    } else {
        // workgroups bid, bid+8, bid+16.. share an XCD (bid % 8); each XCD takes one contiguous run of work items,
        // so the q-blocks of one head -- and the query heads of one GQA group -- stream K/V through one L2, and the
        // longest (causal) blocks of a head are dispatched first.  Measured alternatives (profiles/r1_run28_xcd_map.txt,
        // DESIGN.md 3.1): heads dealt to XCDs in rounds of 8 is 6-14 % slower where it spreads a head's K/V over all
        // eight L2s; shortest-first order -5 %, alternating long/short -21 %.
        int bh, qrank;
        const WorkOrder wo = {CAUSAL ? p.order_group : 0, p.order_fold, p.order_left};      // causal work order: sage_work_order.h
        if (!work_item(wo, bid, pers ? p.nwg : (int)gridDim.x, p.B * p.Hq, nqblk, bh, qrank)) break;
        qblk = nqblk - 1 - qrank;
        b = bh / p.Hq;
        h = bh - b * p.Hq;
        hk = h / p.group;
    }

    // ---- per-sequence geometry ---------------------------------------------------------------
    int Lq = p.Lq, Lk = p.Lk;
    long q_off, k_off, o_off;
    long v_tile0, v_tstride;              // V image index = v_tile0 + t * v_tstride
    const float *qs_ptr, *ks_ptr;
    int qs_stride, ks_tstride;
    if (p.cu_q != nullptr) {              // varlen: packed [sum L, H, D]
        // the prefix arrays are read-only here and the sequence index is wave-uniform: scalar loads, requested together (as vector loads they
        // were a memory round trip of their own behind the work list's; measured neutral at C4, profiles/r4_run_p_attention_phase_trace.txt)
        typedef const __attribute__((address_space(4))) int *cint_p;
        const int bu = __builtin_amdgcn_readfirstlane(b);
        const int q0 = ((cint_p)p.cu_q)[bu], k0 = ((cint_p)p.cu_k)[bu], q1 = ((cint_p)p.cu_q)[bu + 1], k1 = ((cint_p)p.cu_k)[bu + 1];
        const int ks0 = ((cint_p)p.cu_ks)[bu];
        Lq = q1 - q0;
        Lk = k1 - k0;
        if (qblk * BLKQ >= Lq) break;
        q_off = (long)q0 * p.q_sl + (long)h * p.q_sh;
        k_off = (long)k0 * p.k_sl + (long)hk * p.k_sh;
        o_off = (long)q0 * p.o_sl + (long)h * p.o_sh;
        v_tile0 = (long)ks0 * p.Hkv + hk;
        v_tstride = p.Hkv;
        qs_ptr = QF == 0 ? p.q_scale + ((long)((cint_p)p.cu_qs)[bu] + qblk) * p.Hq + h : nullptr;    // [sum nblk, Hq] (fused Q: no stored scales)
        qs_stride = 0;
        ks_ptr = p.k_scale + (long)ks0 * p.Hkv + hk;                  // [sum nblk, Hkv]
        ks_tstride = p.Hkv;
    } else {
        // split-KV (p.kv_split = S > 1): the key range is folded into the kv-head dimension, kv head hk = hk0 * S + chunk and query
        // head h = hk * group + g; the query rows are those of head hk0 * group + g (read in place, no per-chunk copy of Q)
        const int hq = p.kv_split > 1 ? (hk / p.kv_split) * p.group + (h - hk * p.group) : h;
        q_off = (long)b * p.q_sb + (long)hq * p.q_sh;
        k_off = (long)b * p.k_sb + (long)hk * p.k_sh;
        o_off = (long)b * p.o_sb + (long)h * p.o_sh;
        const int ntk = (Lk + BLKK - 1) / BLKK;
        v_tile0 = ((long)b * p.Hkv + hk) * ntk;
        v_tstride = 1;
        qs_ptr = p.q_scale + ((long)b * p.Hq + h) * p.nqs + (long)qblk * p.qs_per_blk;
        qs_stride = 1;
        ks_ptr = p.k_scale + ((long)b * p.Hkv + hk) * p.nks;
        ks_tstride = KTHREAD ? 4 : 1;
    }

    SAGE_TSTAMP(1);
    const int row0 = qblk * BLKQ + wave * 32;        // first query row of this wave
    int my_row = row0 + n;                           // (re-derived behind the pipelined loops, see there)
    // causal mask in the chunk's key coordinates (split-KV: this workgroup sees keys kchunk0 .. kchunk0 + Lk - 1 as 0 .. Lk - 1):
    // key <= row  <=>  local key <= row - kchunk0
    const int kchunk0 = (CAUSAL && p.kv_split > 1 && p.cu_q == nullptr) ? (hk % p.kv_split) * Lk : 0;
    const int crow0 = row0 - kchunk0;
    int cmy_row = my_row - kchunk0;
    const int ntk_all = (Lk + BLKK - 1) / BLKK;      // 64-key images that exist
    int n_iters = (Lk + KT - 1) / KT;
    if (CAUSAL) {
        int lim = (qblk * BLKQ + BLKQ - kchunk0 + KT - 1) / KT;      // <= 0: the whole chunk lies behind the diagonal
        lim = lim > 0 ? lim : 0;
        n_iters = lim < n_iters ? lim : n_iters;
    }

    // ---- Q fragments (B operand of S^T = K Q^T), resident in VGPRs ---------------------------
    v4i qf[C::KSTEPS];
    float qsc;
    // ---- tile staging ------------------------------------------------------------------------
    const unsigned char *kbase = reinterpret_cast<const unsigned char *>(p.k) + k_off;
    static_assert(!VROWS || (!PV_FP8 && MASK == 0), "V rows in place: FP16 PV, unmasked, dense launches");
    // (VROWS: the (batch, kv-head)'s first row; else the image array, indexed by v_tile0 + t * v_tstride)
    const unsigned char *vbase = reinterpret_cast<const unsigned char *>(p.v) + (VROWS ? 2 * ((long)b * p.v_sb + (long)hk * p.v_sh) : 0L);
    constexpr int CPR = D / 16;                                   // 16-B chunks per K row
    // LDS-DMA: every wave-instruction moves 64 x 16 B = 1 KiB; the LDS destination is lane-linear
    // (M0 base + lane*16), so the XOR swizzle of the K image goes on the per-lane SOURCE address.
    // Key rows past Lk are clamped to the last valid row, V images past the last one to the last
    // image (their probabilities are exactly zero: masked scores).
    constexpr int KP = C::K_TILE_BYTES / 1024, VP = C::V_IMG_BYTES / 1024;   // 1-KiB pieces
    // per-lane source offsets are loop-invariant: tile base pointers advance in SGPRs, so a full
    // tile costs no VALU address arithmetic per iteration
    unsigned koff[KP / 4];
#pragma unroll
    for (int i = 0; i < KP / 4; i++) {
        const int e = (wave * (KP / 4) + i) * 64 + lane;       // 16-B slot index inside the tile
        const int row = e / CPR, phys = e % CPR;
        koff[i] = (unsigned)(row * (int)p.k_sl + swz_chunk<D>(row, phys) * 16);
    }
    // VROWS: the tile in LDS is [64 tokens][D halves], at D = 128 with the 64-byte segments of a row XOR-ed by (token & 3) -- the four rows a
    // 32-lane half of a transposing read touches then lie in different banks (unswizzled: 1.5x the read time at two workgroups per CU; D = 64
    // measured no different).  A 1-KiB piece of the DMA is RPP whole rows; the lane's slot inside it is (row l / CPRV, chunk l % CPRV), and
    // since RPP is a multiple of 4 the swizzle is the same for every piece: ONE per-lane source offset, piece bases in SGPRs.
    constexpr int CPRV = D / 8;                                   // 16-B chunks per V row
    constexpr int RPP = 64 / CPRV;                                // rows per 1-KiB piece
    [[maybe_unused]] unsigned voffr = 0;                          // (VROWS) per-lane source offset inside a piece: row * row stride + logical chunk * 16
    if constexpr (VROWS) {
        const int row = lane / CPRV, phys = lane % CPRV;
        const int logical = D == 128 ? (phys ^ ((row & 3) << 2)) : phys;
        voffr = (unsigned)(row * (int)p.v_sl * 2 + logical * 16);
    }
    // the A operand of v_mfma_f32_32x32x16_f16 for channels 32 dt .. + 31 and tokens 16 c .. + 15 of the tile at LDS address `vs`: from the
    // image one ds_read_b128 (lane = channel row of the image); from rows two transposing reads -- the lane ADDRESSES the 8-byte chunk
    // (token 16 c + 8 half + 4 g + (s >> 2), channels 32 dt + 16 hgrp + 4 (s & 3) .. + 3), s = lane & 15, hgrp = (lane >> 4) & 1, and RECEIVES
    // tokens 16 c + 8 half + 4 g + 0 .. 3 of channel 32 dt + (lane & 31): elements 4 half .. 4 half + 3 of the operand, the order P is in
    // (the lane's part of the address is loop-invariant -- one offset per 32-channel tile at D = 128, where the segment swizzle depends on dt -- and
    //  the (c, half) part an immediate: one address add per channel tile and 64-key tile, in the LDS address space so that the offsets fold)
    typedef __attribute__((address_space(3))) unsigned char *lds_bytes;
    [[maybe_unused]] int vr_off[C::DT];
    if constexpr (VROWS) {
        const int s16 = lane & 15, hgrp = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < C::DT; dt++)
            vr_off[dt] = (4 * g + (s16 >> 2)) * (D * 2) + (D == 128 ? (dt ^ (s16 >> 2)) : dt) * 64 + 32 * hgrp + 8 * (s16 & 3);
    }
    auto v_frag = [&](const unsigned char *vs, int dt, int c) -> v4i {
        if constexpr (VROWS) {
            typedef short v4s __attribute__((ext_vector_type(4)));
            typedef short v8s __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) v4s *lds_v4s;
            const lds_bytes a0 = (lds_bytes)vs + vr_off[dt];
            const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a0 + 16 * c * (D * 2)));
            const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a0 + (16 * c + 8) * (D * 2)));
            return __builtin_bit_cast(v4i, (v8s)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
            const int drow = dt * 32 + n;
            return *reinterpret_cast<const v4i *>(vs + drow * 128 + swz_chunk<128>(drow, 4 * g + c) * 16);
        }
    };
    int lane_g = lane;                      // the lane index as the ragged tile loads see it (laundered behind the pipelined loops, see there)
    auto issue_loads = [&](int it, int buf) {
        unsigned char *ks = smem + buf * C::STAGE_BYTES;
        unsigned char *vs = ks + C::K_TILE_BYTES;
        const unsigned char *kt = kbase + (long)it * KT * p.k_sl;
        if (it * KT + KT <= Lk) {
#pragma unroll
            for (int i = 0; i < KP / 4; i++) {
                const int pc = wave * (KP / 4) + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kt + koff[i]),
                                                 (__attribute__((address_space(3))) void *)(ks + pc * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KP / 4; i++) {
                const int pc = wave * (KP / 4) + i;
                const int e = pc * 64 + lane_g;
                const int row = e / CPR, phys = e % CPR;
                int key = it * KT + row;
                key = key < Lk ? key : Lk - 1;
                const unsigned char *src = kbase + (long)key * p.k_sl + swz_chunk<D>(row, phys) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(ks + pc * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            int tv = it * NH + hh;
            tv = tv < ntk_all ? tv : ntk_all - 1;
            if constexpr (VROWS) {          // rows 64 tv .. of the head; rows past Lk are clamped to the last one (their probabilities are exactly zero)
#pragma unroll
                for (int i = 0; i < VP / 4; i++) {
                    const int pc = wave * (VP / 4) + i;
                    int tok = tv * BLKK + pc * RPP + lane_g / CPRV;
                    tok = tok < Lk ? tok : Lk - 1;
                    const int row = pc * RPP + lane_g / CPRV, phys = lane_g % CPRV;
                    const int logical = D == 128 ? (phys ^ ((row & 3) << 2)) : phys;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vbase + (long)tok * p.v_sl * 2 + logical * 16),
                                                     (__attribute__((address_space(3))) void *)(vs + hh * C::V_IMG_BYTES + pc * 1024), 16, 0, 0);
                }
            } else {
            const unsigned char *vt = vbase + (v_tile0 + (long)tv * v_tstride) * (long)C::V_IMG_BYTES;
#pragma unroll
            for (int i = 0; i < VP / 4; i++) {
                const int pc = wave * (VP / 4) + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + pc * 1024 + lane_g * 16),
                                                 (__attribute__((address_space(3))) void *)(vs + hh * C::V_IMG_BYTES + pc * 1024), 16, 0, 0);
            }
            }
        }
    };

    // ---- running state -------------------------------------------------------------------------
    v16f o[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++)
#pragma unroll
        for (int i = 0; i < 16; i++) o[dt][i] = 0.0f;
    float m_run = kNegBig, l_run = 0.0f;
    constexpr float OFF = PV_FP8 ? kFp8Offset : 0.0f;

    // K scales of an iteration are fetched one iteration ahead with SCALAR loads (constant address
    // space, wave-uniform index -> s_load, tracked by lgkmcnt).  An ordinary VMEM load here would be
    // fatal for the pipeline: with LDS-DMA in flight hipcc waits vmcnt(0) at the first use of any
    // VGPR-destination load, draining the in-flight tiles every iteration.
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    const cfloat_p ks_c = (cfloat_p)(ks_ptr);
    float ksc[NH][2];
    auto load_kscales = [&](int it, float (&dst)[NH][2]) {
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            int tk = it * NH + hh;
            tk = tk < ntk_all ? tk : ntk_all - 1;
            const long tb = (long)(tk >> p.ks_shift) * ks_tstride;
            if (KTHREAD) {      // 4 key scales per 64 keys: token%8/2 (quant_per_thread.py:75-83); lane half g uses 2g, 2g+1
                const float s0 = ks_c[tb], s1 = ks_c[tb + 1], s2 = ks_c[tb + 2], s3 = ks_c[tb + 3];
                dst[hh][0] = g ? s2 : s0;
                dst[hh][1] = g ? s3 : s1;
            } else {
                dst[hh][0] = dst[hh][1] = ks_c[tb];
            }
        }
    };
    // LDS ring.  NSTAGE == 3: tiles it+1 and it+2 are in flight while tile it is consumed; a wave waits
    // only for ITS OWN older DMA group with a counted s_waitcnt vmcnt(N) (N = DMA instructions of the
    // younger group) and then meets the others at a raw s_barrier -- __syncthreads() would drain
    // vmcnt(0) and expose the full L2/HBM latency every iteration (cdna_hip_programming.md T3+T4).
    constexpr int NSTAGE = C::NSTAGE;
    constexpr int DMA_PER_TILE = KP / 4 + NH * (VP / 4);          // per wave
    auto ring_wait = [&](bool younger_in_flight) {
        if (younger_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (n_iters > 0) {
        load_kscales(0, ksc);
        issue_loads(0, 0);
    }
    if (n_iters > 1) issue_loads(1, 1);
    // The Q fragments are fetched AFTER the first tiles' LDS-DMA has been issued: hipcc waits vmcnt(0) at the first use of an
    // ordinary VGPR load, and with the Q loads in front it did so after the first DMA instruction -- the Q round trip and the
    // tiles' round trip ran one after the other in every workgroup's prologue.
    if constexpr (QF == 0) {
        const int8_t *qrow = reinterpret_cast<const int8_t *>(p.q) + q_off + (long)my_row * p.q_sl;
        const bool ok = my_row < Lq;
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ks++) {
            v4i z = {0, 0, 0, 0};
            qf[ks] = ok ? *reinterpret_cast<const v4i *>(qrow + 32 * ks + 16 * g) : z;
        }
        // this lane's query-row scale (per-block / per-warp / per-thread granularity)
        int slot;
        const int rin = wave * 32 + n;               // row inside the 128-row block
        if (p.q_gran == QG_PER_BLOCK) slot = 0;
        else if (p.q_gran == QG_PER_WARP32) slot = rin >> 5;
        else if (p.q_gran == QG_PER_WARP16) slot = rin >> 4;
        else if (p.q_gran == QG_PER_THREAD16) slot = (rin >> 4) * 8 + (rin & 7);   // per-thread, WARPQ = 16 (core.py:604,969)
        else slot = (rin >> 5) * 8 + (rin & 7);      // per-thread: quant_per_thread.py:27-37
        qsc = qs_ptr[slot * qs_stride];
    } else {
        // Fused Q quantisation.  The lane holds channels [32 ks + 16 g, +16) of its row for every ks -- the layout of
        // the MFMA B operand -- so it quantises exactly the bytes it needs.  A per-thread group is the rows
        // r, r+8, r+16, r+24 of the wave's 32-row tile, all 128 channels: lanes n = r (mod 8), both halves g.
        constexpr int QDT = (QF == 1 || QF == 3) ? DT_F16 : DT_BF16;
        constexpr bool QBLOCK = QF >= 3;
        const float premul = QBLOCK ? p.q_premul : 1.0f;
        const uint16_t *qrow = reinterpret_cast<const uint16_t *>(p.q) + q_off + (long)my_row * p.q_sl;
        const bool ok = my_row < Lq;
        float x[C::KSTEPS][16];
        float amax = 0.0f;
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ks++) {
            v4u raw[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            if (ok) {
                raw[0] = *reinterpret_cast<const v4u *>(qrow + 32 * ks + 16 * g);
                raw[1] = *reinterpret_cast<const v4u *>(qrow + 32 * ks + 16 * g + 8);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const unsigned w = raw[j >> 3][(j & 7) >> 1];
                float f = ld16<QDT>((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
                if constexpr (QBLOCK) f *= premul;          // x.to(float32) * sm_scale before the abs-max (quant_per_block.py:35-37)
                x[ks][j] = f;
                amax = fmaxf(amax, fabsf(f));
            }
        }
        if constexpr (QBLOCK) {
            // one scale for the workgroup's 128 rows: the wave's maximum, then the four waves' through 16 bytes of LDS (the K / V ring
            // is receiving its first tiles meanwhile; only this word is waited for)
            amax = fmaxf(amax, __shfl_xor(amax, 1));
            amax = fmaxf(amax, __shfl_xor(amax, 2));
            amax = fmaxf(amax, __shfl_xor(amax, 4));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 8));
        amax = fmaxf(amax, __shfl_xor(amax, 16));
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        if constexpr (QBLOCK) {
            __shared__ float q_amax[4];
            if (lane == 0) q_amax[wave] = amax;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            amax = fmaxf(fmaxf(q_amax[0], q_amax[1]), fmaxf(q_amax[2], q_amax[3]));
        }
        const float sc = quant_scale(amax, QBLOCK ? QS_TRITON : QS_TRITON_THREAD);
        const float y = quant_recip(sc);
        qsc = sc;
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ks++) {
            int q8[16];
#pragma unroll
            for (int j = 0; j < 16; j++) q8[j] = QBLOCK ? quant_round_triton(x[ks][j], sc, y) : quant_round_triton_nz(x[ks][j], sc, y);
#pragma unroll
            for (int w = 0; w < 4; w++) qf[ks][w] = (int)pack_int8x4(q8[4 * w], q8[4 * w + 1], q8[4 * w + 2], q8[4 * w + 3]);
        }
    }

    SAGE_TSTAMP(2);
    ring_wait(n_iters > 1);
    SAGE_TSTAMP(3);

    int cur = 0;
    // One K/V tile in the general form: masked, ragged, or one of a workgroup's last two (the whole, unmasked tiles in front of them run the
    // software-pipelined loops below).
    auto tile_iter = [&](const int it) {
        const bool more = (it + 1) < n_iters;
        const bool more2 = (it + 2) < n_iters;
        const int nxt = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        float ksc_next[NH][2];
        if (more) load_kscales(it + 1, ksc_next);
        if (more2) issue_loads(it + 2, (nxt + 1 == NSTAGE) ? 0 : nxt + 1);

        // number of 64-key halves with at least one key this wave may attend to (wave-uniform)
        int nact = 0;
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            const int key0 = it * KT + hh * BLKK;
            if (key0 < Lk && (!CAUSAL || key0 <= crow0 + 31)) nact = hh + 1;
        }
        // ---- attn_mask (Triton-named API only; attn_qk_int8_per_block.py:31-51): additive term per
        //      score in the log2 domain.  bool: 0 / -1e6, and a tile whose whole 128x64 mask block is
        //      False is skipped; float: the mask value itself; out-of-range positions count as False / -1e6.
        float mk[MASK ? NS : 1][16];
        bool skip_tile = false;
        if constexpr (MASK != 0) {
            const unsigned char *mbase = reinterpret_cast<const unsigned char *>(p.mask);
            const long mrow = (long)b * p.m_sb + (long)h * p.m_sh + (long)my_row * p.m_sq;
            int anytrue = 0;
#pragma unroll
            for (int sb = 0; sb < NS; sb++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int key = it * KT + sb * 32 + crow(i, g);
                    const bool inb = (my_row < Lq) && (key < Lk);
                    float add = -1.0e6f;
                    if (inb) {
                        const long idx = mrow + (long)key * p.m_sk;
                        if (MASK == 1) { const bool t = mbase[idx] != 0; add = t ? 0.0f : -1.0e6f; anytrue |= (int)t; }
                        else if (MASK == 2) add = f16_to_f32(reinterpret_cast<const uint16_t *>(mbase)[idx]);
                        else add = bf16_to_f32(reinterpret_cast<const uint16_t *>(mbase)[idx]);
                    }
                    mk[sb][i] = add;
                }
            if (MASK == 1) skip_tile = !__syncthreads_or(anytrue);
        }
        if (nact > 0 && !skip_tile) {
            const unsigned char *ks = smem + cur * C::STAGE_BYTES;
            const unsigned char *vs = ks + C::K_TILE_BYTES;
            const int last_key = it * KT + nact * BLKK - 1;
            const bool full = (MASK == 0) && (nact == NH) && !(CAUSAL && last_key > crow0) && (last_key < Lk);

            // ---- S^T = K Q^T (int8 -> int32), NS sub-tiles of 32 keys ----
            v16i s[NS];
#pragma unroll
            for (int sb = 0; sb < NS; sb++) {
#pragma unroll
                for (int i = 0; i < 16; i++) s[sb][i] = 0;        // sub-tiles past the wave's last key are masked below
                if (sb < 2 * nact) {
                    const int krow = sb * 32 + n;
#pragma unroll
                    for (int kk = 0; kk < C::KSTEPS; kk++) {
                        const v4i a = *reinterpret_cast<const v4i *>(ks + krow * D + swz_chunk<D>(krow, 2 * kk + g) * 16);
                        s[sb] = kk == 0 ? mfma_i8_first(a, qf[kk]) : __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[kk], s[sb], 0, 0, 0);
                    }
                }
            }

            // ---- scales: c multiplies the raw int32 score into the log2 domain, formed in the reference's order
            //      sm_scale*log2e * (q_scale * k_scale)  (qk_int_sv_f8_cuda_sm89.cuh:263-266,334-335) ----
            float cs[NH][2];
#pragma unroll
            for (int hh = 0; hh < NH; hh++) {
                cs[hh][0] = __builtin_ldexpf(p.sm_scale_log2 * (qsc * ksc[hh][0]), kSUnitLog2);
                cs[hh][1] = KTHREAD ? __builtin_ldexpf(p.sm_scale_log2 * (qsc * ksc[hh][1]), kSUnitLog2) : cs[hh][0];
            }

            // ---- online softmax over the iteration's keys ----
            // The row max is taken on the raw int32 scores (c >= 0, so max commutes with the scale);
            // only the per-(half, scale) maxima are converted.  exp2 / row sum / low-precision pack
            // are fused per 8-register chunk so no float copy of S stays live.
            float m_new;
            if (full) {
                float mxc = -INFINITY;
#pragma unroll
                for (int hh = 0; hh < NH; hh++) {
                    int mx0 = INT_MIN, mx1 = INT_MIN;
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            if (KTHREAD && (i & 2)) mx1 = max(mx1, s[2 * hh + u][i]);
                            else mx0 = max(mx0, s[2 * hh + u][i]);
                        }
                    // m_temp = fma(max raw score, scale, -offset)  (attn_utils.cuh:372-384)
                    mxc = fmaxf(mxc, __builtin_fmaf(sfl(mx0), cs[hh][0], -OFF));
                    if (KTHREAD) mxc = fmaxf(mxc, __builtin_fmaf(sfl(mx1), cs[hh][1], -OFF));
                }
                m_new = fmaxf(m_run, pair_max(mxc));
            } else {
                float mx = -INFINITY;
#pragma unroll
                for (int sb = 0; sb < NS; sb++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        if (sb < 2 * nact) {
                            const float cc = cs[sb >> 1][(KTHREAD && (i & 2)) ? 1 : 0];
                            const int key = it * KT + sb * 32 + crow(i, g);
                            const bool ok = (key < Lk) && (!CAUSAL || key <= cmy_row);
                            if constexpr (MASK != 0) mx = fmaxf(mx, (ok ? sfl(s[sb][i]) * cc : 0.0f) + mk[sb][i] - OFF);
                            else mx = fmaxf(mx, ok ? __builtin_fmaf(sfl(s[sb][i]), cc, -OFF) : -INFINITY);
                        }
                    }
                m_new = fmaxf(m_run, pair_max(mx));
            }
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            if (!TWO_LEVEL) {
#pragma unroll
                for (int dt = 0; dt < C::DT; dt++)
#pragma unroll
                    for (int i = 0; i < 16; i++) o[dt][i] *= alpha;
            }

            // P for chunk c (16 keys) of half hh = registers 8u..8u+7 of S^T tile 2hh + (c>>1):
            // exactly the order of the PV B operand (sage_common.h)
            // Row sum.  The reference's FP16-PV CUDA kernels sum the fp16-ROUNDED probabilities (RS_32_to_16, then the tensor-core
            // row sum mma::rowsum_f16f16f32: qk_int_sv_f16_cuda_sm80.cu:313-320, attn_utils.cuh:529-545, DenominatorAccumUnit =
            // kTensorCore in every instantiation); its Triton kernels and FP8 kernels sum the un-rounded ones
            // (attn_qk_int8_per_block.py:57-60, qk_int_sv_f8_cuda_sm90.cu:317-318).  For FP16 PV the C ABI maps the two forms onto
            // the TWO_LEVEL parameter: false = the CUDA kernels' form (SAGE_PV_ACCUM_SINGLE / _TWO_LEVEL: gfx950 accumulates P.V
            // in FP32 whatever tile buffer the reference would use), true = the Triton kernels' form
            // (SAGE_PV_ACCUM_TRITON: tile product folded into the FP32 output, un-rounded denominator).
            constexpr bool sum_rounded = !PV_FP8 && !TWO_LEVEL;
            // FP8 PV, folded score form (SFOLD): as in the pipelined loop, the scale FMA reads the accumulator's bit pattern and subtracts
            // m + bias * c, rounded once per (row, tile, k scale) -- every tile of a launch uses ONE form, which the oracle mirrors
            constexpr bool GFOLD = PV_FP8 && SFOLD && MASK == 0;
            float rs = 0.0f;
            auto p_chunk = [&](auto masked, auto rnd, int hh, int c, float (&e)[8]) {
                const int sb = 2 * hh + (c >> 1), r0 = (c & 1) * 8;
                // (formed per chunk: two FMAs, and nothing more stays live across the chunks -- kept in registers over the whole tile the
                //  pair tipped two D = 64 instantiations one VGPR over their three-waves budget)
                [[maybe_unused]] const float mbf0 = GFOLD ? __builtin_fmaf(__int_as_float(0x3E22F983), cs[hh][0], m_new) : 0.0f;
                [[maybe_unused]] const float mbf1 = (GFOLD && KTHREAD) ? __builtin_fmaf(__int_as_float(0x3E22F983), cs[hh][1], m_new) : mbf0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = r0 + j;
                    const float cc = cs[hh][(KTHREAD && (i & 2)) ? 1 : 0];
                    float v;
                    if constexpr (MASK != 0) {
                        const int key = it * KT + sb * 32 + crow(i, g);
                        v = __builtin_amdgcn_exp2f(((key < Lk) ? sfl(s[sb][i]) * cc : 0.0f) + mk[sb][i] - m_new);
                    } else {
                        if constexpr (GFOLD) v = __builtin_amdgcn_exp2f(__builtin_fmaf(__int_as_float(s[sb][i]), cc, -((KTHREAD && (i & 2)) ? mbf1 : mbf0)));
                        else v = __builtin_amdgcn_exp2f(__builtin_fmaf(sfl(s[sb][i]), cc, -m_new));
                        if constexpr (decltype(masked)::value) {
                            const int key = it * KT + sb * 32 + crow(i, g);
                            const bool ok = (sb < 2 * nact) && (key < Lk) && (!CAUSAL || key <= cmy_row);
                            v = ok ? v : 0.0f;
                        }
                    }
                    e[j] = v;
                    if constexpr (decltype(rnd)::value) rs += (float)(_Float16)v;
                    else rs += v;
                }
            };

            if constexpr (PV_FP8) {
                int pw[NH][8];                   // 32 fp8 per 64-key half = B operand of one K=64 MFMA
                auto build_p = [&](auto masked) {
#pragma unroll
                    for (int hh = 0; hh < NH; hh++)
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            float e[8];
                            p_chunk(masked, std::false_type{}, hh, c, e);
                            int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], __float_as_int(e[0]), false);   // high half is overwritten next
                            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w0, true);
                            int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[4], e[5], __float_as_int(e[4]), false);
                            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[6], e[7], w1, true);
                            pw[hh][2 * c] = w0;
                            pw[hh][2 * c + 1] = w1;
                        }
                };
                if (full) build_p(std::false_type{});
                else build_p(std::true_type{});
                l_run = l_run * alpha + rs;      // lane-partial; the pair is summed in the epilogue
                auto pv = [&](auto fold_tag) {
                    constexpr bool FOLD = decltype(fold_tag)::value;     // tile product from zero, then O = O * alpha + T
#pragma unroll
                    for (int dt = 0; dt < C::DT; dt++) {
                        const int drow = dt * 32 + n;
                        v16f acc;
                        if (FOLD) {
#pragma unroll
                            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
                        } else acc = o[dt];
#pragma unroll
                        for (int hh = 0; hh < NH; hh++) {
                            if (hh < nact) {
                                const unsigned char *vr = vs + hh * C::V_IMG_BYTES + drow * 64;
                                const v4u va = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g) * 16);
                                const v4u vb = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g + 1) * 16);
                                // one K = 64 FP8 MFMA (the plain form: no block scales)
                                const v8i av = {(int)va[0], (int)va[1], (int)va[2], (int)va[3], (int)vb[0], (int)vb[1], (int)vb[2], (int)vb[3]};
                                const v8i bv = {pw[hh][0], pw[hh][1], pw[hh][2], pw[hh][3], pw[hh][4], pw[hh][5], pw[hh][6], pw[hh][7]};
                                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0, 0, 0);    // (zero scale operands: hipcc selects the plain v_mfma_f32_32x32x64_f8f6f4, no scale VGPRs)
                            }
                        }
                        if (FOLD) {
#pragma unroll
                            for (int i = 0; i < 16; i++) o[dt][i] = __builtin_fmaf(o[dt][i], alpha, acc[i]);
                        } else o[dt] = acc;
                    }
                };
                if constexpr (!TWO_LEVEL) pv(std::false_type{});
                else pv(std::true_type{});
            } else {
                v8h pb[NH][4];
                auto build_p = [&](auto masked, auto rnd) {
#pragma unroll
                    for (int hh = 0; hh < NH; hh++)
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            float e[8];
                            p_chunk(masked, rnd, hh, c, e);
#pragma unroll
                            for (int j = 0; j < 8; j++) pb[hh][c][j] = (_Float16)e[j];
                        }
                };
                if (full) build_p(std::false_type{}, std::integral_constant<bool, sum_rounded>{});
                else build_p(std::true_type{}, std::integral_constant<bool, sum_rounded>{});
                l_run = l_run * alpha + rs;
                auto pv = [&](auto fold_tag) {
                    constexpr bool FOLD = decltype(fold_tag)::value;
#pragma unroll
                    for (int dt = 0; dt < C::DT; dt++) {
                        v16f acc;
                        if (FOLD) {
#pragma unroll
                            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
                        } else acc = o[dt];
#pragma unroll
                        for (int hh = 0; hh < NH; hh++) {
                            if (hh < nact) {
#pragma unroll
                                for (int c = 0; c < 4; c++) {
                                    const v8h a = __builtin_bit_cast(v8h, v_frag(vs + hh * C::V_IMG_BYTES, dt, c));
                                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[hh][c], acc, 0, 0, 0);
                                }
                            }
                        }
                        if (FOLD) {
#pragma unroll
                            for (int i = 0; i < 16; i++) o[dt][i] = __builtin_fmaf(o[dt][i], alpha, acc[i]);
                        } else o[dt] = acc;
                    }
                };
                if constexpr (!TWO_LEVEL) pv(std::false_type{});
                else pv(std::true_type{});
            }
        }

        if (more) {
#pragma unroll
            for (int hh = 0; hh < NH; hh++) { ksc[hh][0] = ksc_next[hh][0]; ksc[hh][1] = ksc_next[hh][1]; }
        }
        ring_wait(more2);
        cur = nxt;
    };

    int it = 0;
    if constexpr (MASK == 0) {
        static_assert(NH == 1 && NSTAGE == 3, "the pipelined loops are written for 64-key iterations on the 3-slot ring");
        constexpr bool SIX_BODIES = D == 128 || PV_FP8;         // the pipelined loops' ring slot as a compile-time constant (see the FP8 loop; not D = 64 FP16 PV)
        // whole tiles: it < Lk/64; unmasked for wave 0 (hence all waves): 64 it + 63 <= 128 qblk; two whole tiles follow
        int n_steady = Lk / KT - 2;
        n_steady = n_steady < n_iters - 2 ? n_steady : n_iters - 2;
        if (CAUSAL) {                                  // unmasked for wave 0 (hence all waves): 64 it + 63 <= 128 qblk - kchunk0
            int nd = (qblk * BLKQ - kchunk0) / KT;
            nd = nd > 0 ? nd : 0;
            n_steady = n_steady < nd ? n_steady : nd;
        }
        // DIAG_PIPE (causal): when exactly two tiles follow the steady ones and both are whole, they run through the pipelined body as well (masked
        // there) instead of as general iterations -- also when there is no steady tile at all (the first query block)
        // (FP16 PV: only behind at least one steady tile -- its first body is a form of its own -- and at D = 128: the D = 64 instantiations spill
        //  5-10 VGPRs under their three-waves limit with the two extra bodies)
        constexpr bool DIAG_PIPE = CAUSAL && SAGE_DIAG_PIPE && (PV_FP8 || D == 128);
        const bool diag_ok = DIAG_PIPE && (n_steady > 0 ? n_iters - n_steady == 2 : (PV_FP8 && n_iters == 2 && Lk >= 2 * KT));
        // TAIL_PIPE (non-causal FP8 PV): the two whole tiles the steady loop leaves (it looks two tiles ahead) and a ragged last one behind them take the
        // pipelined body as well -- keys past Lk masked like keys behind the diagonal, the ragged tile requested in the general (clamped) form
        // (FP16 PV, D = 128: the two whole tiles of a call whose Lk is a multiple of 64, behind at least one steady tile -- kinds 1 and 2 as they are)
        constexpr bool TAIL_PIPE = !CAUSAL && SAGE_TAIL_PIPE && (PV_FP8 || D == 128);
        const bool tail_ok = TAIL_PIPE && n_steady == Lk / KT - 2 &&
                             (PV_FP8 ? (n_steady >= 0 && n_iters - n_steady <= (SAGE_TAIL_PIPE == 2 ? 2 : 3)) : (n_steady > 0 && n_iters - n_steady == 2));

        if constexpr (PV_FP8) {
            // ---- software-pipelined steady state -------------------------------------------------------------
            // Iteration t runs softmax(t) on the VALU and deals, between its instruction groups, the PV MFMAs of tile t-1
            // (P and V fragments carried in registers) and the QK^T MFMAs of tile t+1 (K fragments read at the top), so a
            // wave's matrix work is covered by its OWN VALU stream instead of depending on another wave being in the right
            // phase.  The instruction ORDER is the design here, and hipcc re-orders builtin arithmetic freely (it clustered
            // the MFMAs and sank the softmax below them), so every instruction of the main stream is a one-line
            // `asm volatile`: hipcc still allocates the registers, counts its own ds_read / s_load / LDS-DMA and waits for
            // them, but cannot move the statements.  Hazards it therefore does not pad (cdna_hip_programming.md 5.7):
            //   * v_exp_f32 -> first VALU reader: one other instruction in between (groups of two scores are interleaved);
            //   * freshly loaded / written VGPR -> MFMA A/B operand: every MFMA statement opens with s_nop 1;
            //   * MFMA result -> VALU reader: PV results are read at the next iteration's top or after the drain's s_nops,
            //     QK^T results after the 18-instruction tail + barrier; an MFMA taking the previous result whole as C needs none.
            // O is rescaled (rarely) at the top of the next iteration, i.e. after PV(t-1) and before PV(t): the single-level
            // order O = O*alpha + P V, which the FP32 MFMA accumulator makes equivalent to the two-level fold.
            // Ring (3 slots): at the top of iteration t tile t+1 must have landed for every wave (its K is read now), and every
            // wave has finished reading tile t-1, whose slot takes the LDS-DMA of tile t+2.
            // the K = 64 FP8 MFMA without the v_mfma_ld_scale prefix of its block-scaled form (same products; 8 bytes and one VGPR less per MFMA)
            // (no s_nop in front of these MFMAs since round 6 -- 0.7 % of a C3 launch, 24 + 16 nops per FP16 tile: no VALU instruction writes an operand of
            //  theirs within two issue slots -- V / K fragments come from LDS behind the compiler's own waits, P from the previous tile's
            //  conversions, O's rescale lies a barrier and the tile loads away; tools/mfma_hazard_lint.py checks exactly this rule on every listing,
            //  copies the compiler might place in front of a statement included.  SAGE_ABL bit 2 puts the nops back.)
#if SAGE_ABL & 2
#define A_NOP_ "s_nop 1\n\t"
#else
#define A_NOP_ ""
#endif
#define A_PV(acc, av, bv)  asm volatile(A_NOP_ "v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#define A_QK0(acc, a, b)   asm volatile(A_NOP_ "v_mfma_i32_32x32x32_i8 %0, %1, %2, 0x3e22f983" : "=&v"(acc) : "v"(a), "v"(b))
#define A_QK(acc, a, b)    asm volatile(A_NOP_ "v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define A_FENCE()          asm volatile("" ::: "memory")
#if SAGE_ABL & 16
#define A_EXP_ "v_mov_b32"
#else
#define A_EXP_ "v_exp_f32"
#endif
            if (it < n_steady || diag_ok || tail_ok) {
                v16i sA[2], sB[2];
                {
                    const unsigned char *ks0 = smem + cur * C::STAGE_BYTES;
                    v4i kf0[2][C::KSTEPS];
#pragma unroll
                    for (int sb = 0; sb < 2; sb++) {
                        const int krow = sb * 32 + n;
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; kk++)
                            kf0[sb][kk] = *reinterpret_cast<const v4i *>(ks0 + krow * D + swz_chunk<D>(krow, 2 * kk + g) * 16);
                    }
#pragma unroll
                    for (int kk = 0; kk < C::KSTEPS; kk++)
#pragma unroll
                        for (int sb = 0; sb < 2; sb++)
                            sA[sb] = kk == 0 ? mfma_i8_first(kf0[sb][kk], qf[kk]) : __builtin_amdgcn_mfma_i32_32x32x32_i8(kf0[sb][kk], qf[kk], sA[sb], 0, 0, 0);
                }
                v8i pA = {0, 0, 0, 0, 0, 0, 0, 0}, pB = {0, 0, 0, 0, 0, 0, 0, 0};   // P of the previous tile (none yet: zero, the first PV adds nothing)
                v8i vf[C::DT];                                                        // V fragments of the previous tile
#pragma unroll
                for (int dt = 0; dt < C::DT; dt++) vf[dt] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
                static_assert(KP / 4 == VP / 4 && (KP / 4 == 1 || KP / 4 == 2), "asm LDS-DMA: one or two pieces per wave and image");
                const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
                const unsigned voff16 = lane * 16;
                const unsigned koff1m = (KP / 4 == 2) ? koff[KP / 4 - 1] - 1024u : 0u;    // piece 1's source offset minus its inst_offset
                const float sm26 = __builtin_ldexpf(p.sm_scale_log2, kSUnitLog2);
                // the tile's score scales (sm * (q_scale * k_scale)) * 2^26 == (sm * 2^26) * (q_scale * k_scale): exact power-of-two scaling.
                // Carried from iteration to iteration in place of the k scales they are formed from (per-thread k scales are per-lane values:
                // two VGPRs less across the loop; the general iterations behind the loop fetch their k scales again)
                float cs[2];
                cs[0] = sm26 * (qsc * ksc[0][0]);
                cs[1] = KTHREAD ? sm26 * (qsc * ksc[0][1]) : cs[0];
                float alpha_p = 1.0f;            // rescale owed to O before the pending PV (kept out of the iteration's main block)
                [[maybe_unused]] int cmy_row_d = 0;                      // DIAG_PIPE: the lane's row in the chunk's key coordinates, formed behind the loops
                constexpr int kMaskedScore = (int)0xFF000000;           // bit pattern of a score behind the diagonal: below every INT32 score pattern, -1.7e38 as a float
                auto rescale = [&]() {
                    if ((SAGE_ABL & 1) == 0 && __builtin_amdgcn_ballot_w64(alpha_p != 1.0f) != 0) {
#pragma unroll
                        for (int dt = 0; dt < C::DT; dt++)
#pragma unroll
                            for (int i = 0; i < 16; i++) o[dt][i] *= alpha_p;
                    }
                };
                // one tile: sc = scores of tile `it` (complete), sn <- scores of tile it+1, pp = P of tile it-1, pc <- P of tile it
                // (`slot` = the ring slot of tile `it`, a compile-time constant: the six bodies of the loop below are the six combinations of
                //  ring slot and score-register set, so every LDS address of a body is a loop-invariant per-lane offset plus an immediate --
                //  no per-tile address arithmetic on the VALU)
                // `kind` 0: a whole tile of the steady state.  1 / 2 (DIAG_PIPE): the last two tiles of a causal work item in the same instruction order --
                // scores behind the diagonal are replaced by kMaskedScore in front of the row maximum, nothing more is fetched (1: QK^T of the
                // last tile still issued; 2: no next tile at all)
                auto body = [&](auto slot, auto kind, const int n, const int g, v16i (&sc)[2], v16i (&sn)[2], v8i &pp, v8i &pc) {   // (n, g: the lane's row and half, see the remainder loop)
                    constexpr int KIND = decltype(kind)::value;
                    constexpr bool HAS_DMA = KIND == 0 || KIND == 3, HAS_NEXT = KIND != 2, DIAG = KIND == 1 || KIND == 2;       // (3, TAIL_PIPE: the tile requested is ragged)
                    rescale();
                    const int CUR = slot;            // (std::integral_constant in the six-body loop: folds; an int in the remainder loop)
                    const int nxt = (CUR + 1 == NSTAGE) ? 0 : CUR + 1, nn = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
                    const unsigned char *vs = smem + CUR * C::STAGE_BYTES + C::K_TILE_BYTES;
                    const unsigned char *ksn = smem + nxt * C::STAGE_BYTES;
                    if (SAGE_ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (bit 32: timing probe, the tile is not waited for)
                    if ((SAGE_ABL & 4) == 0) __builtin_amdgcn_s_barrier();
                    if constexpr (HAS_DMA) {
                        // K: this wave's KP/4 pieces (1 KiB each, swizzled through the per-lane source offset); V: its VP/4 pieces.
                        // inst_offset advances the global and the LDS address together, so piece 1 reuses piece 0's M0.
                        // (KIND 3: piece 1's clamped offset minus its inst_offset can be negative, and the VGPR offset of the SGPR-base form is unsigned:
                        //  base 1 KiB down, offsets 1 KiB up)
                        const unsigned char *ktp = kbase + (long)(it + 2) * KT * p.k_sl - (KIND == 3 ? 1024 : 0);
                        const unsigned char *vtp = vbase + (v_tile0 + (long)(it + 2) * v_tstride) * (long)C::V_IMG_BYTES + wave * (VP / 4) * 1024;
                        const unsigned ldk = lds_base + nn * C::STAGE_BYTES + wave * (KP / 4) * 1024;
                        const unsigned ldv = lds_base + nn * C::STAGE_BYTES + C::K_TILE_BYTES + wave * (VP / 4) * 1024;
                        unsigned keep;
                        // (KIND 3: tile it + 2 is the ragged one -- its rows past Lk are read from the last row there is, as issue_loads does; the V image
                        //  is whole, zero-padded)
                        unsigned k0 = koff[0], k1m = koff1m;
                        if constexpr (KIND == 3) {
                            const int rmax = Lk - 1 - (it + 2) * KT, l64 = g * 32 + n;
                            const int r0 = ((wave * (KP / 4)) * 64 + l64) / CPR, r1 = ((wave * (KP / 4) + 1) * 64 + l64) / CPR;
                            k0 = k0 + 1024u - (unsigned)((r0 > rmax ? r0 - rmax : 0) * (int)p.k_sl);
                            if constexpr (KP / 4 == 2) k1m = k1m + 1024u - (unsigned)((r1 > rmax ? r1 - rmax : 0) * (int)p.k_sl);
                        }
                        if constexpr (KP / 4 == 2)
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                                         "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %7, %4\n\tglobal_load_lds_dwordx4 %7, %4 offset:1024\n\t"
                                         "s_mov_b32 m0, %0"
                                         : "=&s"(keep) : "v"(k0), "v"(k1m), "s"(ktp), "s"(vtp), "s"(ldk), "s"(ldv), "v"(voff16) : "memory");
                        else
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %1, %2\n\t"
                                         "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %6, %3\n\t"
                                         "s_mov_b32 m0, %0"
                                         : "=&s"(keep) : "v"(k0), "s"(ktp), "s"(vtp), "s"(ldk), "s"(ldv), "v"(voff16) : "memory");
                    }

                    // ---- PV(t-1) MFMAs 0, 1; row maximum of S(t) (plain code: it only has to finish before the first exponential) ----
                    // (nothing is in flight on lgkmcnt here, so hipcc's own wait for the V fragments in front of this MFMA is free;
                    //  the K-fragment reads and the scalar load of the next K scales are issued behind it)
                    A_PV(o[0], vf[0], pp);
                    A_FENCE();
                    // the next tile's k scales (scalar loads).  Per-thread groups: the lane halves take different pairs of the tile's four scales;
                    // the products with the lane's q scale are formed under EXEC instead of selecting first (SAGE_KSEL: three VALU
                    // instructions per tile less than move + select + multiply; same multiplications in the same order: same bits)
                    float ksc_next[NH][2];
                    [[maybe_unused]] float ks4[4];
                    if constexpr (!HAS_NEXT) {
                    } else if constexpr (KTHREAD && SAGE_KSEL) {
                        const long tb = (long)((it + 1) >> p.ks_shift) * ks_tstride;      // (whole tiles only in this loop: it + 1 < ntk_all)
                        ks4[0] = ks_c[tb]; ks4[1] = ks_c[tb + 1]; ks4[2] = ks_c[tb + 2]; ks4[3] = ks_c[tb + 3];
                    } else load_kscales(it + 1, ksc_next);
                    v4i kfa[C::KSTEPS], kfb[C::KSTEPS];
                    if constexpr (HAS_NEXT) {
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; kk++) {
                            kfa[kk] = *reinterpret_cast<const v4i *>(ksn + n * D + swz_chunk<D>(n, 2 * kk + g) * 16);
                        }
                    }
                    A_FENCE();
                    // (the tile's score scale in a masked tile: never below 2^-100, so that kMaskedScore * c is a large negative number even when a
                    //  q or k scale is zero -- c < 2^-100 multiplies scores below 2^-5 into nothing against any m and any offset either way: same bits)
                    if constexpr (DIAG) {
                        if (CAUSAL ? (crow0 < it * KT + KT - 1) : (Lk < it * KT + KT)) {     // (wave-uniform: a wave whose first row sees the whole tile has nothing to mask)
                            // the lane's last visible key of this tile (causal: its row; otherwise the last key there is), minus its half's offset
                            const int x = (CAUSAL ? cmy_row_d : Lk - 1) - it * KT - 4 * g;
#pragma unroll
                            for (int u = 0; u < 2; u++)
#pragma unroll
                                for (int i = 0; i < 16; i++) sc[u][i] = (u * 32 + 8 * (i >> 2) + (i & 3) <= x) ? sc[u][i] : kMaskedScore;
                        }
                        cs[0] = fmaxf(cs[0], 0x1p-100f);
                        cs[1] = KTHREAD ? fmaxf(cs[1], 0x1p-100f) : cs[0];
                    }
                    int mx0 = INT_MIN, mx1 = INT_MIN;
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int i = ((SAGE_ABL & 8) ? 14 : 0); i < 16; i++) {
                            if (KTHREAD && (i & 2)) mx1 = max(mx1, sc[u][i]);
                            else mx0 = max(mx0, sc[u][i]);
                        }
                    float mxc = __builtin_fmaf(sfl(mx0), cs[0], -OFF);
                    if (KTHREAD) mxc = fmaxf(mxc, __builtin_fmaf(sfl(mx1), cs[1], -OFF));
                    const float m_new = fmaxf(m_run, pair_max(mxc));
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    // SFOLD: what the scale FMA subtracts is the row maximum plus the bias of the score's bit pattern in this tile's scale
                    const float mb0 = SFOLD ? __builtin_fmaf(__int_as_float(0x3E22F983), cs[0], m_new) : m_new;
                    const float mb1 = (SFOLD && KTHREAD) ? __builtin_fmaf(__int_as_float(0x3E22F983), cs[1], m_new) : mb0;
                    if constexpr (C::DT > 1) A_PV(o[1], vf[1], pp);
                    A_FENCE();
                    if constexpr (HAS_NEXT) {
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; kk++) {
                            kfb[kk] = *reinterpret_cast<const v4i *>(ksn + (32 + n) * D + swz_chunk<D>(32 + n, 2 * kk + g) * 16);
                        }
                    }
                    A_FENCE();

                    // ---- exponentials / row sum / fp8 pack in 16 groups of two scores ----
                    float rs0, rs1;                  // partial row sums: defined by the first group (grp(0) / g4c(0))
                    auto grp = [&](int h) {          // scores 2h, 2h+1 of the lane's 32, in PV operand order: one statement =
                        const int c = h >> 2, j0 = (h & 3) * 2;                  // 2 x (bias sub, scale fma, exp2, row-sum add) + fp8 pack
                        const int sb = c >> 1, i0 = (c & 1) * 8 + j0;
                        float t0, t1;
                        const float ca = cs[(KTHREAD && (i0 & 2)) ? 1 : 0], cb = cs[(KTHREAD && ((i0 + 1) & 2)) ? 1 : 0];
#define SAGE_GRP(SCALE2, PACK)                                                                                                  \
                        asm volatile(SCALE2("%2", "%3", "%5", "%6", "%7", "%8", "%9")                                             \
                                     "v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"                                                  \
                                     "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t" PACK                                      \
                                     : "+v"(rs0), "+v"(rs1), "=&v"(t0), "=&v"(t1), "+v"(pc[h >> 1])                               \
                                     : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(ca), "v"(cb), "v"((KTHREAD && (i0 & 2)) ? mb1 : mb0))
                        // (the first group DEFINES the two partial row sums -- 0 + p is p: no zero initialisation, no add)
#define SAGE_GRP0(SCALE2)                                                                                                       \
                        asm volatile(SCALE2("%0", "%1", "%3", "%4", "%5", "%6", "%7")                                             \
                                     "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\ts_nop 0\n\tv_cvt_pk_fp8_f32 %2, %0, %1"                \
                                     : "=&v"(rs0), "=&v"(rs1), "+v"(pc[0])                                                        \
                                     : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(ca), "v"(cb), "v"((KTHREAD && (i0 & 2)) ? mb1 : mb0))
                        if constexpr (SFOLD) {
                            if (h == 0) SAGE_GRP0(SAGE_SCALE2_FOLD);
                            else if ((h & 1) == 0) SAGE_GRP(SAGE_SCALE2_FOLD, "v_cvt_pk_fp8_f32 %4, %2, %3");
                            else SAGE_GRP(SAGE_SCALE2_FOLD, "v_cvt_pk_fp8_f32 %4, %2, %3 op_sel:[0,0,1]");
                        } else {
                            if (h == 0) SAGE_GRP0(SAGE_SCALE2_EXACT);
                            else if ((h & 1) == 0) SAGE_GRP(SAGE_SCALE2_EXACT, "v_cvt_pk_fp8_f32 %4, %2, %3");
                            else SAGE_GRP(SAGE_SCALE2_EXACT, "v_cvt_pk_fp8_f32 %4, %2, %3 op_sel:[0,0,1]");
                        }
#undef SAGE_GRP0
#undef SAGE_GRP
                    };
                    auto &qfr = qf;                  // (named in the generic body itself: a lambda nested in it does not capture through it otherwise)
                    auto qk_next = [&](int sb, int kk) {
                        if constexpr (!HAS_NEXT) return;
                        else if (kk == 0) A_QK0(sn[sb], (sb == 0 ? kfa[0] : kfb[0]), qfr[0]);
                        else A_QK(sn[sb], (sb == 0 ? kfa[kk] : kfb[kk]), qfr[kk]);
                    };
                    auto read_v = [&](int dt) {      // V fragments of THIS tile for the next iteration's PV
                        const int drow = dt * 32 + n;
                        const unsigned char *vr = vs + drow * 64;
                        const v4u a = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g) * 16);
                        const v4u b = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g + 1) * 16);
                        vf[dt] = v8i{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
                    };
                    // D = 128: four scores per statement (four independent chains instead of two), in three parts -- scale (bias add + FMA), the
                    // exponentials, row sum + fp8 pack -- so that every MFMA sits directly in front of a group's four exponentials: the
                    // quarter-rate instructions overlap a running MFMA best (tools/microbench/ubench5), the next group's scale part follows
                    // them, then this group's sums and packs (two sets of temporaries).  Against MFMAs in front of the scale parts:
                    // +0.6 % at N = 32k, +1.2 % at C3 non-causal, bit-identical (profiles/r6_run_i_loop_trim_ab.txt).
                    float ua[4], ub[4];
                    auto g4s = [&](int w, float (&u)[4]) {
                        const int sb = w >> 2, i0 = 4 * (w & 3);
                        if constexpr (SFOLD)
                            asm volatile(SAGE_SCALE2_FOLD("%0", "%1", "%4", "%5", "%8", "%8", "%10") SAGE_SCALE2_FOLD("%2", "%3", "%6", "%7", "%9", "%9", "%11")
                                         : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                                         : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(sc[sb][i0 + 2]), "v"(sc[sb][i0 + 3]), "v"(cs[0]), "v"(cs[1]), "v"(mb0), "v"(mb1));
                        else
                            asm volatile("v_add_f32 %0, 0xbe22f983, %4\n\tv_add_f32 %1, 0xbe22f983, %5\n\t"
                                         "v_add_f32 %2, 0xbe22f983, %6\n\tv_add_f32 %3, 0xbe22f983, %7\n\t"
                                         "v_fma_f32 %0, %0, %8, -%10\n\tv_fma_f32 %1, %1, %8, -%10\n\t"
                                         "v_fma_f32 %2, %2, %9, -%11\n\tv_fma_f32 %3, %3, %9, -%11"
                                         : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                                         : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(sc[sb][i0 + 2]), "v"(sc[sb][i0 + 3]), "v"(cs[0]), "v"(cs[1]), "v"(mb0), "v"(mb1));
                    };
                    auto g4e = [&](float (&u)[4]) {
                        asm volatile(A_EXP_ " %0, %0\n\t" A_EXP_ " %1, %1\n\t" A_EXP_ " %2, %2\n\t" A_EXP_ " %3, %3" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
                    };
                    auto g4c = [&](int w, float (&u)[4]) {
                        if (w == 0)
                            asm volatile("v_add_f32 %0, %3, %5\n\tv_add_f32 %1, %4, %6\n\t"
                                         "v_cvt_pk_fp8_f32 %2, %3, %4\n\tv_cvt_pk_fp8_f32 %2, %5, %6 op_sel:[0,0,1]"
                                         : "=&v"(rs0), "=&v"(rs1), "+v"(pc[w]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                        else
                            asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\t"
                                         "v_cvt_pk_fp8_f32 %2, %3, %4\n\tv_cvt_pk_fp8_f32 %2, %5, %6 op_sel:[0,0,1]"
                                         : "+v"(rs0), "+v"(rs1), "+v"(pc[w]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                    };
                    if constexpr (C::DT == 4) {
                        g4s(0, ua);
                        A_PV(o[2], vf[2], pp);   g4e(ua); g4s(1, ub); g4c(0, ua);
                        A_PV(o[3], vf[3], pp);   g4e(ub); g4s(2, ua); g4c(1, ub);
                        qk_next(0, 0);           g4e(ua); g4s(3, ub); g4c(2, ua);
                        qk_next(0, 1);           g4e(ub); g4s(4, ua); g4c(3, ub);
                        qk_next(0, 2);           g4e(ua);
                        A_FENCE(); read_v(0); read_v(1); A_FENCE();
                        g4s(5, ub); g4c(4, ua);
                        qk_next(0, 3);           g4e(ub); g4s(6, ua); g4c(5, ub);
                        qk_next(1, 0);           g4e(ua); g4s(7, ub); g4c(6, ua);
                        qk_next(1, 1);           g4e(ub);
                        qk_next(1, 2);           g4c(7, ub);
                        qk_next(1, 3);
                        A_FENCE(); read_v(2); read_v(3); A_FENCE();
                    } else {                         // D = 64: two PV MFMAs (dealt above), four QK^T MFMAs
                        grp(0); grp(1); grp(2); grp(3);
                        qk_next(0, 0); grp(4); grp(5); grp(6);
                        qk_next(0, 1); grp(7); grp(8); grp(9);
                        A_FENCE(); read_v(0); A_FENCE();
                        qk_next(1, 0); grp(10); grp(11); grp(12);
                        qk_next(1, 1);
                        A_FENCE(); read_v(1); A_FENCE();
                        grp(13); grp(14); grp(15);
                    }
                    l_run = l_run * alpha + (rs0 + rs1);
                    if constexpr (!HAS_NEXT) {
                    } else if constexpr (KTHREAD && SAGE_KSEL) {
                        unsigned long long keep;
                        asm volatile("v_mul_f32 %0, %3, %7\n\tv_mul_f32 %1, %4, %7\n\t"
                                     "s_mov_b64 %2, exec\n\ts_mov_b64 exec, %8\n\t"
                                     "v_mul_f32 %0, %5, %7\n\tv_mul_f32 %1, %6, %7\n\t"
                                     "s_mov_b64 exec, %2\n\t"
                                     "v_mul_f32 %0, %9, %0\n\tv_mul_f32 %1, %9, %1"
                                     : "=&v"(cs[0]), "=&v"(cs[1]), "=&s"(keep)
                                     : "s"(ks4[0]), "s"(ks4[1]), "s"(ks4[2]), "s"(ks4[3]), "v"(qsc), "s"(0xFFFFFFFF00000000ull), "v"(sm26));
                    } else {
                        cs[0] = sm26 * (qsc * ksc_next[0][0]);
                        cs[1] = KTHREAD ? sm26 * (qsc * ksc_next[0][1]) : cs[0];
                    }
                    alpha_p = alpha;
                    it++;
                };
                // The loop enters with tile `it` in slot 0 (cur == 0: no general iteration runs in front of it) and its scores in set A.  Six bodies --
                // the six combinations of ring slot and register set, the slot a compile-time constant in each -- bring both back to where they
                // were; what is left of the count (< 6) runs one body at a time on a run-time slot, renamed B -> A behind it (a few times per workgroup).
                {
                    static_assert(SIX_BODIES, "FP8 PV: both head sizes run the six-body loop");
                    const int left = n_steady - it;
                    int n6 = left / 6, r = left - 6 * n6;
                    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
#pragma nounroll
                    for (; n6 > 0; n6--) {
                        body(I0{}, I0{}, n, g, sA, sB, pA, pB); body(I1{}, I0{}, n, g, sB, sA, pB, pA); body(I2{}, I0{}, n, g, sA, sB, pA, pB);
                        body(I0{}, I0{}, n, g, sB, sA, pB, pA); body(I1{}, I0{}, n, g, sA, sB, pA, pB); body(I2{}, I0{}, n, g, sB, sA, pB, pA);
                    }
                    // (the remainder body's per-lane LDS offsets are derived behind the six-body loop from a lane index the compiler cannot see
                    //  through: formed in front of it they would stay live across it, next to that loop's own -- registers D = 64 does not have)
                    int lane_r;                      // (v_mbcnt again rather than a copy of `lane`: nothing of the thread index has to live through the loop)
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_r));
                    const int n_r = lane_r & 31, g_r = lane_r >> 5;
#pragma nounroll
                    for (; r > 0; r--) {
                        body(cur, I0{}, n_r, g_r, sA, sB, pA, pB);
                        SAGE_RENAME_S();
                        pA = pB;
                        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                    }
                    // Causal: the work item's last two tiles (whole tiles both when exactly two are left: n_steady <= Lk / 64 - 2) keep the pipeline's
                    // order instead of draining it into two general iterations -- the scores of the first are in set A already, the last steady
                    // body requested the second.  Same arithmetic per score as a general tile's (bias subtraction + FMA against the same m): same bits.
                    // (the last bodies' per-lane offsets from a lane index of their own: shared with the remainder loop's they stay live across it)
                    int lane_t;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
                    const int n_t = lane_t & 31, g_t = lane_t >> 5;
                    if constexpr (DIAG_PIPE) {
                        if (diag_ok) {
                            cmy_row_d = row0 - kchunk0 + n_t;
                            body(cur, I1{}, n_t, g_t, sA, sB, pA, pB);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                            body(cur, I2{}, n_t, g_t, sB, sA, pB, pA);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                        }
                    }
                    if constexpr (TAIL_PIPE) {
                        if (tail_ok) {
                            if (SAGE_TAIL_PIPE != 2 && n_iters - it == 3) {           // a ragged tile behind the two whole ones: this body requests it
                                body(cur, std::integral_constant<int, 3>{}, n_t, g_t, sA, sB, pA, pB);
                                SAGE_RENAME_S();
                                pA = pB;
                                cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                            }
                            body(cur, I1{}, n_t, g_t, sA, sB, pA, pB);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                            body(cur, I2{}, n_t, g_t, sB, sA, pB, pA);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                        }
                    }
                    n = n_t;                         // (what follows -- the drain's k scales, the general tiles -- reads the lane's row and half formed behind the loop)
                    g = g_t;
                }
                // drain: PV of the last pipelined tile; then every wave must be past its V reads before the general
                // iteration issues the LDS-DMA of tile it+2 into that slot
                rescale();
#pragma unroll
                for (int dt = 0; dt < C::DT; dt++) A_PV(o[dt], vf[dt], pA);
                if (it < n_iters) load_kscales(it, ksc);           // (the loop carried the products, not the k scales: the general iterations start from these)
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" : "+v"(sA[0]), "+v"(sA[1])::"memory");   // (sA as operand: its readers stay below)
                __builtin_amdgcn_s_barrier();
            }
#undef A_PV
#undef A_QK0
#undef A_QK
#undef A_FENCE
#undef A_NOP_
#undef A_EXP_
        } else {
            // ---- software-pipelined steady state, FP16 PV --------------------------------------------------------------------
            // Same structure as the FP8 loop above; differences:
            //  * PV(t-1) is 4 x DT v_mfma_f32_32x32x16_f16 whose V fragments do not fit in registers next to two score tiles,
            //    so they are read from LDS as they are needed, one 32-channel tile (4 x ds_read_b128) ahead of its MFMAs;
            //  * V(t-1) must therefore stay in LDS through iteration t.  The 3-slot ring still suffices because a slot's K and
            //    V regions are filled separately: at the top of iteration t the LDS-DMA brings K(t+2) into the K region of
            //    slot (t+2)%3 (K(t-1), read in iteration t-2, is dead) and V(t+1) into the V region of slot (t+1)%3 (V(t-2),
            //    read in iteration t-1, is dead); K(t+1) and V(t-1) were requested one and two iterations ago.
            //  * FP16 PV rounds P to 2^-11, so the running maximum the exponent is taken against need not be the true one: m_run is a
            //    REFERENCE that is refreshed (and O, l rescaled) only when a row of the wave has a score more than kLazyTau above it -- P <= 2^kLazyTau
            //    fits fp16 with its full mantissa, small probabilities keep more of theirs, and softmax is invariant to the reference.  The
            //    reference's kernels update m and rescale every tile (attn_utils.cuh:394-431); on random data a wave then rescales its 64 O
            //    registers in 60-85 % of the tiles of a C2 block (some row of 32 sets a record), here once or twice per work item.  FP8 PV
            //    cannot do this: e4m3's 2^-4 rounding of P is re-rolled by any change of the reference.
            //    The scores themselves take the exact form fma(s, c, -m) (SAGE_SCALE2_EXACT), as the reference's (attn_utils.cuh:445-449).
            //    Measured (profiles/r6_run_a_fp16_exact_lazy_ab.txt): exact scores cost the FP16 routes 4-5 % against round 5's folded bias, the lazy
            //    reference returns it (C2 -0.8 %, C4 causal +1.2 % against round 5; +4.2 % / +4.5 % against exact scores refreshed on every move).
#define SAGE_SCALE2 SAGE_SCALE2_EXACT
#if SAGE_ABL & 2
#define A_NOP_ "s_nop 1\n\t"
#else
#define A_NOP_ ""                   // (see the FP8 loop)
#endif
#define A_PV16(acc, av, bv) asm volatile(A_NOP_ "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#define A_QK0(acc, a, b)   asm volatile(A_NOP_ "v_mfma_i32_32x32x32_i8 %0, %1, %2, 0x3e22f983" : "=&v"(acc) : "v"(a), "v"(b))
#define A_QK(acc, a, b)    asm volatile(A_NOP_ "v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// ... with the nop: a work item's FIRST body, whose MFMAs read the zeros O and P start from -- the compiler materialises those where they are
// first used, right in front of the asm MFMA (the lint's VALU-write rule found them) -- and every body of the D = 64 two-body form, whose
// first body is a run-time case
#define A_PV16N(acc, av, bv) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#define A_QK0N(acc, a, b)   asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 0x3e22f983" : "=&v"(acc) : "v"(a), "v"(b))
#define A_QKN(acc, a, b)    asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define A_FENCE()          asm volatile("" ::: "memory")
            if (it < n_steady) {                 // (diag_ok / tail_ok need a steady tile here: n_steady > 0)
                v16i sA[2], sB[2];
                {
                    const unsigned char *ks0 = smem + cur * C::STAGE_BYTES;
                    v4i kf0[2][C::KSTEPS];
#pragma unroll
                    for (int sb = 0; sb < 2; sb++) {
                        const int krow = sb * 32 + n;
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; kk++)
                            kf0[sb][kk] = *reinterpret_cast<const v4i *>(ks0 + krow * D + swz_chunk<D>(krow, 2 * kk + g) * 16);
                    }
#pragma unroll
                    for (int kk = 0; kk < C::KSTEPS; kk++)
#pragma unroll
                        for (int sb = 0; sb < 2; sb++)
                            sA[sb] = kk == 0 ? mfma_i8_first(kf0[sb][kk], qf[kk]) : __builtin_amdgcn_mfma_i32_32x32x32_i8(kf0[sb][kk], qf[kk], sA[sb], 0, 0, 0);
                }
                v4i pA[4], pB[4];                  // P of a tile as fp16 pairs: [chunk of 16 keys][word] = B operands of the PV MFMAs
#pragma unroll
                for (int c = 0; c < 4; c++) { pA[c] = v4i{0, 0, 0, 0}; pB[c] = v4i{0, 0, 0, 0}; }
                const float sm26 = __builtin_ldexpf(p.sm_scale_log2, kSUnitLog2);
                static_assert(KP / 4 == 1 || KP / 4 == 2, "asm LDS-DMA: one or two K pieces per wave");
                static_assert(VP / 4 == 2 * (KP / 4), "fp16 V image = two K tiles");
                const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
                [[maybe_unused]] const unsigned voff16 = lane * 16;
                const unsigned koff1m = (KP / 4 == 2) ? koff[KP / 4 - 1] - 1024u : 0u;
                constexpr float kLazyTau = 8.0f;
                // per-thread k scale groups: the tile's four scales stay scalars from body to body and their products with the lane's q scale are
                // formed under EXEC (SAGE_KSEL, see the FP8 loop); `ksc` is restored from them behind the loop for the general tiles
                [[maybe_unused]] float ks4c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (KTHREAD && SAGE_KSEL) {
                    const long tb = (long)(it >> p.ks_shift) * ks_tstride;
                    ks4c[0] = ks_c[tb]; ks4c[1] = ks_c[tb + 1]; ks4c[2] = ks_c[tb + 2]; ks4c[3] = ks_c[tb + 3];
                }
                float alpha_p = 1.0f;
                bool moved_p = false;              // wave-uniform: the previous tile refreshed the reference, O owes alpha_p
                auto rescale = [&]() {
                    if (moved_p) {
#pragma unroll
                        for (int dt = 0; dt < C::DT; dt++)
#pragma unroll
                            for (int i = 0; i < 16; i++) o[dt][i] *= alpha_p;
                    }
                };
                // CUDA kernel form (TWO_LEVEL false): row sum of the fp16-rounded P; Triton kernel form (TWO_LEVEL true): of the
                // un-rounded P (see tile_iter)
                constexpr bool RSUM16 = !TWO_LEVEL;
                // (`slot` = the ring slot of tile `it`, a compile-time constant as in the FP8 loop: every LDS address of a body is a loop-invariant
                //  per-lane offset plus an immediate.  `first`: the work item's first body has no previous tile -- P = 0 against the (finite) V
                //  of the current slot instead of slot (cur + 2) % 3, which nothing has been written to yet)
                //  `kind` 0: a steady tile; 1 / 2 (DIAG_PIPE): a causal work item's last two tiles, masked in front of the row maximum -- 1 requests only
                //  V(t+1) and still issues the QK^T of the last tile, 2 fetches nothing and has no next tile)
                [[maybe_unused]] int cmy_row_d = 0;
                constexpr int kMaskedScore = (int)0xFF000000;           // (see the FP8 loop)
                auto body = [&](auto slot, auto first, auto kind, v16i (&sc)[2], v16i (&sn)[2], v4i (&pp)[4], v4i (&pc)[4]) {
                    constexpr int KIND = decltype(kind)::value;
                    constexpr bool HAS_NEXT = KIND != 2, DIAG = KIND != 0;
                    rescale();
                    const int CUR = slot;            // (std::integral_constant in the six-body loop: folds; an int in the remainder loop)
                    const int nxt = (CUR + 1 == NSTAGE) ? 0 : CUR + 1, nn = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
                    const bool FIRST = first;        // (std::true_type / false_type in the D = 128 loops; a bool in the D = 64 loop)
                    const int prv = FIRST ? CUR : nn;                                      // slot of tile t-1 = (cur + 2) % 3
                    const unsigned char *vsp = smem + prv * C::STAGE_BYTES + C::K_TILE_BYTES;
                    const unsigned char *ksn = smem + nxt * C::STAGE_BYTES;
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if constexpr (KIND == 1) {       // V(t+1) alone (the drain's form)
                        unsigned char *vsn = smem + nxt * C::STAGE_BYTES + C::K_TILE_BYTES;
                        [[maybe_unused]] const unsigned char *vt = vbase + (VROWS ? 0L : (v_tile0 + (long)(it + 1) * v_tstride) * (long)C::V_IMG_BYTES);
#pragma unroll
                        for (int i = 0; i < VP / 4; i++) {
                            const int pc_ = wave * (VP / 4) + i;
                            if constexpr (VROWS)
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vbase + ((long)(it + 1) * BLKK + pc_ * RPP) * p.v_sl * 2 + voffr),
                                                                 (__attribute__((address_space(3))) void *)(vsn + pc_ * 1024), 16, 0, 0);
                            else
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + pc_ * 1024 + lane * 16),
                                                                 (__attribute__((address_space(3))) void *)(vsn + pc_ * 1024), 16, 0, 0);
                        }
                    } else if constexpr (KIND == 0) {   // LDS-DMA: K(t+2) -> K region of slot nn, V(t+1) -> V region of slot nxt (SGPR-base form, see the FP8 loop)
                        const unsigned char *ktp = kbase + (long)(it + 2) * KT * p.k_sl;
                        const unsigned ldk = lds_base + nn * C::STAGE_BYTES + wave * (KP / 4) * 1024;
                        const unsigned ldv = lds_base + nxt * C::STAGE_BYTES + C::K_TILE_BYTES + wave * (VP / 4) * 1024;
                        unsigned keep;
                        if constexpr (VROWS) {
                            // V rows: piece i of the wave = rows (wave * VP / 4 + i) * RPP .. of tile it + 1; inst_offset advances the LDS address by
                            // 1 KiB per piece and the global address with it, so every piece's SGPR base is its rows' address minus 1024 i
                            const long ps = (long)RPP * p.v_sl * 2;
                            const unsigned char *v0 = vbase + (long)(it + 1) * BLKK * p.v_sl * 2 + (long)wave * (VP / 4) * ps;
                            const unsigned char *v1 = v0 + (ps - 1024);
                            if constexpr (KP / 4 == 2) {
                                const unsigned char *v2 = v1 + (ps - 1024), *v3 = v2 + (ps - 1024);
                                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                                             "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                                             "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                                             "global_load_lds_dwordx4 %6, %7\n\tglobal_load_lds_dwordx4 %6, %8 offset:1024\n\t"
                                             "global_load_lds_dwordx4 %6, %9 offset:2048\n\tglobal_load_lds_dwordx4 %6, %10 offset:3072\n\t"
                                             "s_mov_b32 m0, %0"
                                             : "=&s"(keep) : "v"(koff[0]), "v"(koff1m), "s"(ktp), "s"(ldk), "s"(ldv), "v"(voffr), "s"(v0), "s"(v1), "s"(v2), "s"(v3) : "memory");
                            } else {
                                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                                             "global_load_lds_dwordx4 %1, %2\n\t"
                                             "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                                             "global_load_lds_dwordx4 %5, %6\n\tglobal_load_lds_dwordx4 %5, %7 offset:1024\n\t"
                                             "s_mov_b32 m0, %0"
                                             : "=&s"(keep) : "v"(koff[0]), "s"(ktp), "s"(ldk), "s"(ldv), "v"(voffr), "s"(v0), "s"(v1) : "memory");
                            }
                        } else {
                        const unsigned char *vtp = vbase + (v_tile0 + (long)(it + 1) * v_tstride) * (long)C::V_IMG_BYTES + wave * (VP / 4) * 1024;
                        if constexpr (KP / 4 == 2)
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                                         "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %7, %4\n\tglobal_load_lds_dwordx4 %7, %4 offset:1024\n\t"
                                         "global_load_lds_dwordx4 %7, %4 offset:2048\n\tglobal_load_lds_dwordx4 %7, %4 offset:3072\n\t"
                                         "s_mov_b32 m0, %0"
                                         : "=&s"(keep) : "v"(koff[0]), "v"(koff1m), "s"(ktp), "s"(vtp), "s"(ldk), "s"(ldv), "v"(voff16) : "memory");
                        else
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %1, %2\n\t"
                                         "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                                         "global_load_lds_dwordx4 %6, %3\n\tglobal_load_lds_dwordx4 %6, %3 offset:1024\n\t"
                                         "s_mov_b32 m0, %0"
                                         : "=&s"(keep) : "v"(koff[0]), "s"(ktp), "s"(vtp), "s"(ldk), "s"(ldv), "v"(voff16) : "memory");
                        }
                    }
                    float cs[2];
                    if constexpr (KTHREAD && SAGE_KSEL) {
                        unsigned long long keep;
                        asm volatile("v_mul_f32 %0, %3, %7\n\tv_mul_f32 %1, %4, %7\n\t"
                                     "s_mov_b64 %2, exec\n\ts_mov_b64 exec, %8\n\t"
                                     "v_mul_f32 %0, %5, %7\n\tv_mul_f32 %1, %6, %7\n\t"
                                     "s_mov_b64 exec, %2\n\t"
                                     "v_mul_f32 %0, %9, %0\n\tv_mul_f32 %1, %9, %1"
                                     : "=&v"(cs[0]), "=&v"(cs[1]), "=&s"(keep)
                                     : "s"(ks4c[0]), "s"(ks4c[1]), "s"(ks4c[2]), "s"(ks4c[3]), "v"(qsc), "s"(0xFFFFFFFF00000000ull), "v"(sm26));
                    } else {
                        cs[0] = sm26 * (qsc * ksc[0][0]);
                        cs[1] = KTHREAD ? sm26 * (qsc * ksc[0][1]) : cs[0];
                    }
                    // V fragments of tile t-1, one 32-channel tile at a time (two register sets, alternating)
                    v4i vfa[4], vfb[4];
                    auto read_v = [&](int dt, v4i (&vf)[4]) {
#pragma unroll
                        for (int c = 0; c < 4; c++) vf[c] = v_frag(vsp, dt, c);
                    };
                    read_v(0, vfa);
                    A_FENCE();
                    if constexpr (DIAG && CAUSAL) {      // (the score scale of a masked tile never below 2^-100: see the FP8 loop; non-causal: whole tiles only)
                        if (crow0 < it * KT + KT - 1) {
                            const int x = cmy_row_d - it * KT - 4 * g;
#pragma unroll
                            for (int u = 0; u < 2; u++)
#pragma unroll
                                for (int i = 0; i < 16; i++) sc[u][i] = (u * 32 + 8 * (i >> 2) + (i & 3) <= x) ? sc[u][i] : kMaskedScore;
                        }
                        cs[0] = fmaxf(cs[0], 0x1p-100f);
                        cs[1] = KTHREAD ? fmaxf(cs[1], 0x1p-100f) : cs[0];
                    }
                    // ---- row maximum of S(t) (plain code) ----
                    int mx0 = INT_MIN, mx1 = INT_MIN;
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            if (KTHREAD && (i & 2)) mx1 = max(mx1, sc[u][i]);
                            else mx0 = max(mx0, sc[u][i]);
                        }
                    float mxc = __builtin_fmaf(sfl(mx0), cs[0], -OFF);
                    if (KTHREAD) mxc = fmaxf(mxc, __builtin_fmaf(sfl(mx1), cs[1], -OFF));
                    const float m_t = pair_max(mxc);
                    const bool moved = __builtin_amdgcn_ballot_w64(m_t > m_run + kLazyTau) != 0;       // some row of the wave left the window
                    const float m_new = moved ? fmaxf(m_run, m_t) : m_run;
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // (1.0 where the reference stays)
                    m_run = m_new;
                    const float mb0 = m_new, mb1 = m_new;
                    A_FENCE();
                    if constexpr (C::DT > 1) read_v(1, vfb);
                    A_FENCE();

                    float rs0, rs1;                  // partial row sums: defined by grp(0)
                    auto grp = [&](int h) {          // scores 2h, 2h+1 of the lane's 32: bias sub, scale fma, exp2, fp16 pack, row sum
                        const int c = h >> 2, j0 = (h & 3) * 2;
                        const int sb = c >> 1, i0 = (c & 1) * 8 + j0;
                        float t0, t1;
                        const float ca = cs[(KTHREAD && (i0 & 2)) ? 1 : 0], cb = cs[(KTHREAD && ((i0 + 1) & 2)) ? 1 : 0];
                        const float mb = (KTHREAD && (i0 & 2)) ? mb1 : mb0;       // (i0 is even: both scores share the k scale)
                        if (h == 0) {
                            // the first group DEFINES the two partial row sums (0 + p is p: no zero initialisation; the un-rounded form needs no add)
                            if constexpr (RSUM16)
                                asm volatile(SAGE_SCALE2("%2", "%3", "%5", "%6", "%7", "%8", "%9")
                                             "v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                                             "s_nop 0\n\tv_cvt_pk_f16_f32 %4, %2, %3\n\t"
                                             "v_fma_mix_f32 %0, %4, 1.0, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                                             "v_fma_mix_f32 %1, %4, 1.0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                             : "=&v"(rs0), "=&v"(rs1), "=&v"(t0), "=&v"(t1), "=&v"(pc[c][h & 3])
                                             : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(ca), "v"(cb), "v"(mb));
                            else
                                asm volatile(SAGE_SCALE2("%0", "%1", "%3", "%4", "%5", "%6", "%7")
                                             "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\ts_nop 0\n\tv_cvt_pk_f16_f32 %2, %0, %1"
                                             : "=&v"(rs0), "=&v"(rs1), "=&v"(pc[c][h & 3])
                                             : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(ca), "v"(cb), "v"(mb));
                        } else if constexpr (RSUM16) {
                            // row sum of the ROUNDED pair in FP32: v_fma_mix_f32 reads a half of the packed word as its f16 operand
                            // (rs += f32(half) * 1.0).  Not v_dot2_f32_f16: the dot instructions flush fp16 subnormals whatever the
                            // mode, and a long row's many probabilities below 2^-14 are a visible share of its denominator (seen as
                            // outputs 0.6-1.7 % too large on Lk = 333 with per-block scales).  The reference takes this sum from the
                            // tensor core (see tile_iter).
                            asm volatile(SAGE_SCALE2("%2", "%3", "%5", "%6", "%7", "%8", "%9")
                                         "v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                                         "s_nop 0\n\tv_cvt_pk_f16_f32 %4, %2, %3\n\t"
                                         "v_fma_mix_f32 %0, %4, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                                         "v_fma_mix_f32 %1, %4, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                         : "+v"(rs0), "+v"(rs1), "=&v"(t0), "=&v"(t1), "=&v"(pc[c][h & 3])
                                         : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(ca), "v"(cb), "v"(mb));
                        } else {
                            asm volatile(SAGE_SCALE2("%2", "%3", "%5", "%6", "%7", "%8", "%9")
                                         "v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                                         "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
                                         "v_cvt_pk_f16_f32 %4, %2, %3"
                                         : "+v"(rs0), "+v"(rs1), "=&v"(t0), "=&v"(t1), "=v"(pc[c][h & 3])
                                         : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(ca), "v"(cb), "v"(mb));
                        }
                    };
                    v4i kfa[C::KSTEPS], kfb[C::KSTEPS];
                    auto read_k = [&](int sb, v4i (&kf)[C::KSTEPS]) {
                        if constexpr (!HAS_NEXT) return;
                        const int krow = sb * 32 + n;
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; kk++)
                            kf[kk] = *reinterpret_cast<const v4i *>(ksn + krow * D + swz_chunk<D>(krow, 2 * kk + g) * 16);
                    };
                    auto &o_r = o;                   // (named in the generic body itself, as qfr below)
                    constexpr bool NOPS = !SIX_BODIES || std::is_same<std::decay_t<decltype(first)>, std::true_type>::value;
                    auto pv4 = [&](int dt, v4i (&vf)[4], int c) {
                        if constexpr (NOPS) A_PV16N(o_r[dt], vf[c], pp[c]);
                        else A_PV16(o_r[dt], vf[c], pp[c]);
                    };
                    auto &qfr = qf;                  // (named in the generic body itself: a lambda nested in it does not capture through it otherwise)
                    auto qk_next = [&](int sb, int kk) {
                        if constexpr (!HAS_NEXT) return;
                        else if constexpr (NOPS) {
                            if (kk == 0) A_QK0N(sn[sb], (sb == 0 ? kfa[0] : kfb[0]), qfr[0]);
                            else A_QKN(sn[sb], (sb == 0 ? kfa[kk] : kfb[kk]), qfr[kk]);
                        } else {
                            if (kk == 0) A_QK0(sn[sb], (sb == 0 ? kfa[0] : kfb[0]), qfr[0]);
                            else A_QK(sn[sb], (sb == 0 ? kfa[kk] : kfb[kk]), qfr[kk]);
                        }
                    };
                    // D = 128: four scores per statement in three parts (scale / exponentials / pack + row sum), as in the FP8 loop: three MFMAs
                    // per group, the first directly in front of the group's exponentials, and no nop between exponential and pack.  16 PV +
                    // 8 QK^T MFMAs (32 cycles each); two 32-channel tiles are in flight and their MFMAs alternate, so consecutive MFMAs never share
                    // an accumulator (a dependent MFMA issued behind other instructions waits for the full write-back).  Against round 5's
                    // two-score groups with the MFMAs in front of them: C2 +2.5 ... +3.9 %, Triton-named API +2 ... +3.7 %, C4 +2.1 / +2.2 %,
                    // bit-identical (profiles/r6_run_i_loop_trim_ab.txt).
                    [[maybe_unused]] float ua[4], ub[4];
                    auto g4s = [&](int w, float (&u)[4]) {
                        const int sb = w >> 2, i0 = 4 * (w & 3);
                        asm volatile("v_add_f32 %0, 0xbe22f983, %4\n\tv_add_f32 %1, 0xbe22f983, %5\n\t"
                                     "v_add_f32 %2, 0xbe22f983, %6\n\tv_add_f32 %3, 0xbe22f983, %7\n\t"
                                     "v_fma_f32 %0, %0, %8, -%10\n\tv_fma_f32 %1, %1, %8, -%10\n\t"
                                     "v_fma_f32 %2, %2, %9, -%10\n\tv_fma_f32 %3, %3, %9, -%10"
                                     : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                                     : "v"(sc[sb][i0]), "v"(sc[sb][i0 + 1]), "v"(sc[sb][i0 + 2]), "v"(sc[sb][i0 + 3]), "v"(cs[0]), "v"(cs[1]), "v"(mb0));
                    };
                    auto g4e = [&](float (&u)[4]) {
                        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
                    };
                    auto g4c = [&](int w, float (&u)[4]) {
                        if constexpr (RSUM16) {
                            if (w == 0)
                                asm volatile("v_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %6, %7\n\t"
                                             "v_fma_mix_f32 %0, %2, 1.0, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, 1.0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                             "v_fma_mix_f32 %0, %3, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %3, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                             : "=&v"(rs0), "=&v"(rs1), "=&v"(pc[w >> 1][(2 * w) & 3]), "=&v"(pc[w >> 1][(2 * w + 1) & 3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                            else
                                asm volatile("v_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %6, %7\n\t"
                                             "v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                             "v_fma_mix_f32 %0, %3, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %3, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                             : "+v"(rs0), "+v"(rs1), "=&v"(pc[w >> 1][(2 * w) & 3]), "=&v"(pc[w >> 1][(2 * w + 1) & 3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                        } else {
                            if (w == 0)
                                asm volatile("v_add_f32 %0, %4, %6\n\tv_add_f32 %1, %5, %7\n\t"
                                             "v_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %6, %7"
                                             : "=&v"(rs0), "=&v"(rs1), "=&v"(pc[w >> 1][(2 * w) & 3]), "=&v"(pc[w >> 1][(2 * w + 1) & 3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                            else
                                asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %7\n\t"
                                             "v_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %6, %7"
                                             : "+v"(rs0), "+v"(rs1), "=&v"(pc[w >> 1][(2 * w) & 3]), "=&v"(pc[w >> 1][(2 * w + 1) & 3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
                        }
                    };
                    if constexpr (C::DT == 4) {
                        g4s(0, ua);
                        pv4(0, vfa, 0); g4e(ua); pv4(1, vfb, 0); g4s(1, ub); pv4(0, vfa, 1); g4c(0, ua);
                        pv4(1, vfb, 1); g4e(ub); pv4(0, vfa, 2); g4s(2, ua); pv4(1, vfb, 2); g4c(1, ub);
                        pv4(0, vfa, 3);
                        A_FENCE(); read_v(2, vfa); A_FENCE();
                        g4e(ua);
                        pv4(1, vfb, 3);
                        A_FENCE(); read_v(3, vfb); A_FENCE();
                        g4s(3, ub); pv4(2, vfa, 0); g4c(2, ua);
                        pv4(3, vfb, 0); g4e(ub); pv4(2, vfa, 1); g4s(4, ua); pv4(3, vfb, 1); g4c(3, ub);
                        pv4(2, vfa, 2); g4e(ua); pv4(3, vfb, 2); g4s(5, ub);
                        pv4(2, vfa, 3);
                        A_FENCE(); read_k(0, kfa); A_FENCE();
                        g4c(4, ua);
                        pv4(3, vfb, 3);
                        A_FENCE(); read_k(1, kfb); A_FENCE();
                        g4e(ub); g4s(6, ua);
                        qk_next(0, 0); g4c(5, ub);
                        qk_next(1, 0); g4e(ua); qk_next(0, 1); g4s(7, ub); qk_next(1, 1); g4c(6, ua);
                        qk_next(0, 2); g4e(ub); qk_next(1, 2); g4c(7, ub);
                        qk_next(0, 3); qk_next(1, 3);
                    } else {                         // D = 64: 8 PV + 4 QK^T MFMAs
                        pv4(0, vfa, 0); grp(0);
                        pv4(0, vfa, 1); grp(1);
                        pv4(0, vfa, 2); grp(2);
                        pv4(0, vfa, 3); grp(3);
                        A_FENCE(); read_k(0, kfa); A_FENCE();
                        pv4(1, vfb, 0); grp(4);
                        pv4(1, vfb, 1); grp(5);
                        pv4(1, vfb, 2); grp(6);
                        pv4(1, vfb, 3); grp(7);
                        A_FENCE(); read_k(1, kfb); A_FENCE();
                        grp(8); grp(9);
                        qk_next(0, 0); grp(10); grp(11);
                        qk_next(0, 1); grp(12);
                        qk_next(1, 0); grp(13); grp(14);
                        qk_next(1, 1); grp(15);
                    }
                    A_FENCE();
                    if constexpr (!HAS_NEXT) {
                    } else if constexpr (KTHREAD && SAGE_KSEL) {
                        const long tb = (long)((it + 1) >> p.ks_shift) * ks_tstride;      // (scalar loads, consumed at the next top; tile it + 1 exists: two whole tiles follow the loop)
                        ks4c[0] = ks_c[tb]; ks4c[1] = ks_c[tb + 1]; ks4c[2] = ks_c[tb + 2]; ks4c[3] = ks_c[tb + 3];
                    } else {
                        float ksc_next[NH][2];
                        load_kscales(it + 1, ksc_next);          // scalar load, consumed at the next top (behind the drained lgkmcnt)
                        ksc[0][0] = ksc_next[0][0];
                        ksc[0][1] = ksc_next[0][1];
                    }
                    l_run = l_run * alpha + (rs0 + rs1);
                    alpha_p = alpha;
                    moved_p = moved;
                    it++;
                };
                // The loop enters with tile `it` in slot 0 (cur == 0) and its scores in set A.  The first body is peeled (it alone has no previous
                // tile) and renamed B -> A; six bodies -- the six combinations of ring slot and register set from slot 1 on, the slot a compile-time
                // constant in each -- bring both back to where they were; what is left of the count (< 6) runs one body at a time on a run-time
                // slot, renamed behind it (a few times per workgroup).
                {
#define SAGE_RENAME() do { SAGE_RENAME_S(); _Pragma("unroll") for (int c = 0; c < 4; c++) pA[c] = pB[c]; } while (0)
                    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                    if constexpr (SIX_BODIES) {
                        const int left = n_steady - it - 1;
                        body(I0{}, std::true_type{}, I0{}, sA, sB, pA, pB);
                        SAGE_RENAME();
                        cur = 1;
                        int n6 = left / 6, r = left - 6 * n6;
#pragma nounroll
                        for (; n6 > 0; n6--) {
                            body(I1{}, std::false_type{}, I0{}, sA, sB, pA, pB); body(I2{}, std::false_type{}, I0{}, sB, sA, pB, pA); body(I0{}, std::false_type{}, I0{}, sA, sB, pA, pB);
                            body(I1{}, std::false_type{}, I0{}, sB, sA, pB, pA); body(I2{}, std::false_type{}, I0{}, sA, sB, pA, pB); body(I0{}, std::false_type{}, I0{}, sB, sA, pB, pA);
                        }
#pragma nounroll
                        for (; r > 0; r--) {
                            body(cur, std::false_type{}, I0{}, sA, sB, pA, pB);
                            SAGE_RENAME();
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                        }
                    } else {            // (D = 64 FP16 PV: the non-causal forms have no registers to spare under the three-waves limit for six bodies -- the run-time-slot body twice)
                        bool first_rt = true;
                        if ((n_steady - it) & 1) {           // odd count: one tile first, renamed (once per workgroup)
                            body(cur, first_rt, I0{}, sA, sB, pA, pB);
                            first_rt = false;
                            SAGE_RENAME();
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                        }
#pragma nounroll
                        while (it < n_steady) {
                            body(cur, first_rt, I0{}, sA, sB, pA, pB);
                            first_rt = false;
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                            body(cur, std::false_type{}, I0{}, sB, sA, pB, pA);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                        }
                    }
                    // Causal: the last two tiles keep the pipeline's order (see the FP8 loop); scores of the first in set A, K of the second requested
                    if constexpr (DIAG_PIPE || TAIL_PIPE) {
                        if (diag_ok || tail_ok) {
                            cmy_row_d = crow0 + n;
                            body(cur, std::false_type{}, I1{}, sA, sB, pA, pB);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                            body(cur, std::false_type{}, I2{}, sB, sA, pB, pA);
                            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                        }
                    }
#undef SAGE_RENAME
                }
                if constexpr (KTHREAD && SAGE_KSEL) { ksc[0][0] = g ? ks4c[2] : ks4c[0]; ksc[0][1] = g ? ks4c[3] : ks4c[1]; }
                // drain: PV of the last pipelined tile (its V is in slot (cur + 2) % 3); V(it+1) is requested so that the general
                // iteration finds tile it+1 "in flight" as a whole; then tile `it` must be complete and every wave past its reads
                rescale();
                {
                    const int nxt = (cur + 1 == NSTAGE) ? 0 : cur + 1;
                    const int prv = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
                    const unsigned char *vsp = smem + prv * C::STAGE_BYTES + C::K_TILE_BYTES;
                    unsigned char *vsn = smem + nxt * C::STAGE_BYTES + C::K_TILE_BYTES;
                    [[maybe_unused]] const unsigned char *vt = vbase + (VROWS ? 0L : (v_tile0 + (long)(it + 1) * v_tstride) * (long)C::V_IMG_BYTES);
                    // V(it+1) lands in the V region of slot nxt, which held V(it-2): the tile whose fragments the LAST loop
                    // iteration read (late in its body: channel tiles 2, 3).  Every wave must be past those reads before any
                    // wave's DMA may overwrite them -- inside the loop the barrier at the top of the body orders this; here
                    // nothing did, and a fast wave could corrupt a slow wave's last PV (seen as 32 rows x channels 64..127 of
                    // one head differing between two identical calls, once in a few hundred launches).
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (!(diag_ok || tail_ok))           // (behind the last bodies nothing is left to request)
#pragma unroll
                    for (int i = 0; i < VP / 4; i++) {
                        const int pc_ = wave * (VP / 4) + i;
                        if constexpr (VROWS)        // (tile it + 1 is whole: the pipelined loop ends two whole tiles before the last)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vbase + ((long)(it + 1) * BLKK + pc_ * RPP) * p.v_sl * 2 + voffr),
                                                             (__attribute__((address_space(3))) void *)(vsn + pc_ * 1024), 16, 0, 0);
                        else
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + pc_ * 1024 + lane * 16),
                                                             (__attribute__((address_space(3))) void *)(vsn + pc_ * 1024), 16, 0, 0);
                    }
#pragma unroll
                    for (int dt = 0; dt < C::DT; dt++) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const v4i a = v_frag(vsp, dt, c);
                            A_PV16(o[dt], a, pA[c]);
                        }
                    }
                    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" : "+v"(sA[0]), "+v"(sA[1])::"memory");   // (sA as operand: its readers stay below)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VP / 4) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
#undef SAGE_SCALE2
#undef A_PV16
#undef A_QK0
#undef A_QK
#undef A_FENCE
#undef A_NOP_
#undef A_PV16N
#undef A_QK0N
#undef A_QKN
        }
    }
    if constexpr (MASK == 0) {
        // the general iterations' and the epilogue's per-lane values (LDS offsets, the lane's row) are re-derived here from a lane index the
        // compiler cannot see through: formed before the pipelined loops they stay live across them, and the D = 64 instantiations, which
        // have no register to spare under their three-waves limit, spill them
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_g));
        n = lane_g & 31;
        g = lane_g >> 5;
        my_row = row0 + n;
        cmy_row = my_row - kchunk0;
    }
#pragma nounroll
    for (; it < n_iters; it++) tile_iter(it);
    SAGE_TSTAMP(4);
    __syncthreads();      // (raw barriers above do not order the epilogue's LDS reuse against stray waits)
    SAGE_TSTAMP(5);
    // persistent launch: the next ticket is requested here, behind the last tile, and read after the output rows are on their way
    if (pers && !own_empty) {
        if (wave == 0 && lane_g == 0) next_k_v = __hip_atomic_fetch_add(kpl()->sched + 32 * my_queue(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        have_next = true;
    }

    // ---- epilogue: normalise, (x v_scale, + v_mean), cast, transpose through LDS, store rows ----
    __builtin_amdgcn_s_setreg(kHwregModeFp16Ovfl, 0);
    const float l_tot = pair_sum(l_run);
    const float inv = l_tot > 0.0f ? __builtin_amdgcn_rcpf(l_tot) : 0.0f;
    if (p.lse != nullptr && g == 0 && my_row < Lq) {
        long lidx = (p.cu_q != nullptr) ? ((long)h * p.lse_sh + p.cu_q[b] + my_row)
                                        : ((long)b * p.Hq + h) * (long)p.Lq + my_row;
        p.lse[lidx] = __builtin_amdgcn_logf(l_tot) + m_run;   // v_log_f32 is log2
    }
    // all waves are past the last tile barrier: the staging LDS is free
    unsigned char *obuf = smem + wave * (32 * D * 2);
    // per-channel epilogue factors, fetched per 32-wide d tile as straight-line batches of 16-byte
    // vectors (a per-element "load if non-null" makes hipcc branch around every load and wait
    // vmcnt(0) each time: 128 serial L2 round trips per workgroup)
    const float *vsc = PV_FP8 ? p.v_scale + ((long)b * p.Hkv + hk) * D : nullptr;
    const float *vmn = (p.v_mean != nullptr) ? p.v_mean + ((long)b * p.Hkv + hk) * D : nullptr;
    // every factor of the tile is requested before the first one is used: one exposed memory latency per workgroup
    // instead of one per 32-channel tile (the slot is idle for the co-resident workgroup's sake until this one retires)
    v4f sc4[C::DT][4], mn4[C::DT][4];
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const v4f one = {1.0f, 1.0f, 1.0f, 1.0f};
            sc4[dt][r4] = PV_FP8 ? *reinterpret_cast<const v4f *>(vsc + dt * 32 + 8 * r4 + 4 * g) : one;
        }
        if (vmn != nullptr) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) mn4[dt][r4] = *reinterpret_cast<const v4f *>(vmn + dt * 32 + 8 * r4 + 4 * g);
        } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) { const v4f z = {0.0f, 0.0f, 0.0f, 0.0f}; mn4[dt][r4] = z; }
        }
    }
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int d0 = dt * 32 + 8 * r4 + 4 * g;           // 4 consecutive d: regs 4*r4 .. 4*r4+3
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[j] = o[dt][4 * r4 + j] * inv;
                if (PV_FP8) x[j] *= sc4[dt][r4][j];
                x[j] += mn4[dt][r4][j];
            }
            v2u pk;
            if (p.out_dtype == DT_F16) {
                pk[0] = (unsigned)f32_to_f16_rne(x[0]) | ((unsigned)f32_to_f16_rne(x[1]) << 16);
                pk[1] = (unsigned)f32_to_f16_rne(x[2]) | ((unsigned)f32_to_f16_rne(x[3]) << 16);
            } else {
                pk[0] = (unsigned)f32_to_bf16_rne(x[0]) | ((unsigned)f32_to_bf16_rne(x[1]) << 16);
                pk[1] = (unsigned)f32_to_bf16_rne(x[2]) | ((unsigned)f32_to_bf16_rne(x[3]) << 16);
            }
            const int q8 = d0 >> 2;                             // 8-byte chunk index in the row
            const int Q = (q8 >> 1) ^ (n & 7);                 // 16-B chunk, XOR-swizzled by row
            *reinterpret_cast<v2u *>(obuf + n * (D * 2) + Q * 16 + (q8 & 1) * 8) = pk;
        }
    }
    // each wave transposes through its OWN 32-row region: its ds_writes and ds_reads execute in order, no workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        constexpr int LPR = D * 2 / 16;          // lanes per row (16 B each)
        constexpr int RPP = 64 / LPR;            // rows per pass
        unsigned char *obase = reinterpret_cast<unsigned char *>(p.o) + 2 * o_off;
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; pass++) {
            const int r = pass * RPP + lane_g / LPR, Q = lane_g % LPR;
            const v4u val = *reinterpret_cast<const v4u *>(obuf + r * (D * 2) + (Q ^ (r & 7)) * 16);
            const int grow = row0 + r;
            if (grow < Lq) *reinterpret_cast<v4u *>(obase + 2 * ((long)grow * p.o_sl) + Q * 16) = val;
        }
    }
#if SAGE_ATTN_TRACE
    SAGE_TSTAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SAGE_TSTAMP(7);
    if (wave == 0 && p.trace != nullptr && bid < p.trace_wgs) {
        if (lane < 8) p.trace[16 * bid + lane] = ttrace[lane];
        if (lane == 9) p.trace[16 * bid + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        if (lane == 10) p.trace[16 * bid + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        if (lane == 11) p.trace[16 * bid + 11] = blockIdx.x;
        if (lane == 12) p.trace[16 * bid + 12] = (unsigned)qblk;
    }
#endif
    } while (0);
    if (!pers) break;
    // (the barrier also separates this item's LDS transposes from the next item's first tiles)
    if (wave_s == 0) { const int t = resolve_ticket(); if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) s_ticket[tpar] = t; }
    __syncthreads();
    bid = __builtin_amdgcn_readfirstlane(s_ticket[tpar]);      // (wave-uniform: everything derived from it stays in SGPRs)
    tpar ^= 1;
    }
    // A persistent launch leaves its counter block as it found it: every workgroup checks out once it has no ticket left (all its ticket
    // atomics have returned by then -- their values were consumed), and the last one to leave writes the zeros, so the caller can hand the
    // same block to the next launch of the stream without a memset in between (round 6; the Python layer keeps one block per stream).
    if constexpr (PERS_OK) {
        if (pers && wave_s == 0) {
            unsigned *const sched = kpl()->sched;
            const int lane_x = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            unsigned gone = 0;
            if (lane_x == 0) gone = __hip_atomic_fetch_add(sched + kAttnSchedDoneWord, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)__builtin_amdgcn_readfirstlane(gone) + 1u == gridDim.x) {
                if (lane_x < 32) __hip_atomic_store(sched + 32 * lane_x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane_x == 32) __hip_atomic_store(sched + kAttnSchedDoneWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace sage
