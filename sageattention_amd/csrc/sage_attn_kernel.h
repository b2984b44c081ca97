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
//    per trip: ring slot and score-register set are compile-time constants in each) -- the fragments
//    sage_attn_loop_f8.h / sage_attn_loop_f16.h, included inside the kernel body.  A work item's last tiles
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
#include "sage_attn_loop_f8.h"          // (the FP8-PV pipelined loop, its last-tile bodies and its drain: a fragment of this function)
        } else {
#include "sage_attn_loop_f16.h"         // (the FP16-PV pipelined loop)
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
        if (lane < 8 || lane == 8 || lane >= 13) p.trace[16 * bid + lane] = ttrace[lane];
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
