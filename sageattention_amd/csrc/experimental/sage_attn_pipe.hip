// EXPERIMENT, NOT BUILT INTO libsage_gfx950.so (kept for the round-2 scheduling work; DESIGN.md ladder).
// sage_attn_pipe.hip -- software-pipelined variant of the FP8-PV attention kernel (see ../sage_attn.hip for
// the design): QK^T of tile t+1 is issued inside the softmax basic block of tile t, 4-stage LDS ring,
// 2 waves/SIMD.  Measured (profiles/r1_run19_variants_pipe.txt): bit-identical output, but 244-256 VGPRs with
// spills under hipcc's scheduling: 1068 vs 1233 TFLOP/s for the 3-waves/SIMD kernel at C3 -> not adopted.
#include "../sage_common.h"
#include "../sage_kernels.h"
#include <climits>
#include <type_traits>

#define SAGE_GLDS 1
#define SAGE_MXPV 1

namespace sage {

template <int D> struct PipeCfg {
    static constexpr int KT = BLKK;
    static constexpr int K_TILE_BYTES = KT * D;
    static constexpr int V_IMG_BYTES = D * 64;
    static constexpr int STAGE_BYTES = K_TILE_BYTES + V_IMG_BYTES;
    static constexpr int O_BYTES = BLKQ * D * 2;
    static constexpr int LDS_BYTES = (4 * STAGE_BYTES > O_BYTES) ? 4 * STAGE_BYTES : O_BYTES;
    static constexpr int KSTEPS = D / 32;
    static constexpr int DT = D / 32;
};

__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

template <int D, bool CAUSAL, bool KTHREAD, bool TWO_LEVEL>
__global__ void __launch_bounds__(256, 2)
sage_attn_pipe_kernel(const AttnParams p)
{
    constexpr bool PV_FP8 = true;
    constexpr int NH = 1, MASK = 0;
    using C = PipeCfg<D>;
    constexpr int KT = C::KT;
    constexpr int NS = 2 * NH;                       // 32-key S^T sub-tiles per iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;      // query row inside the wave's 32-row tile
    const int g = lane >> 5;      // k-group (operand half)

    // ---- work item: XCD-aware, heavy-first --------------------------------------------------
    // blocks b, b+8, b+16.. share an XCD (b % 8); give each XCD a contiguous run of work items
    // so that the q-blocks of one (batch, kv-head) hit the same L2.
    const int nwg = gridDim.x;
    int wid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        wid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int nqblk = p.nqblk;
    const int bh = wid / nqblk;
    const int qblk = nqblk - 1 - (wid - bh * nqblk);   // longest (causal) blocks first
    const int b = bh / p.Hq;
    const int h = bh - b * p.Hq;
    const int hk = h / p.group;

    // ---- per-sequence geometry ---------------------------------------------------------------
    int Lq = p.Lq, Lk = p.Lk;
    long q_off, k_off, o_off;
    long v_tile0, v_tstride;              // V image index = v_tile0 + t * v_tstride
    const float *qs_ptr, *ks_ptr;
    int qs_stride, ks_tstride;
    if (p.cu_q != nullptr) {              // varlen: packed [sum L, H, D]
        const int q0 = p.cu_q[b], k0 = p.cu_k[b];
        Lq = p.cu_q[b + 1] - q0;
        Lk = p.cu_k[b + 1] - k0;
        if (qblk * BLKQ >= Lq) return;
        q_off = (long)q0 * p.q_sl + (long)h * p.q_sh;
        k_off = (long)k0 * p.k_sl + (long)hk * p.k_sh;
        o_off = (long)q0 * p.o_sl + (long)h * p.o_sh;
        v_tile0 = (long)p.cu_ks[b] * p.Hkv + hk;
        v_tstride = p.Hkv;
        qs_ptr = p.q_scale + ((long)p.cu_qs[b] + qblk) * p.Hq + h;    // [sum nblk, Hq]
        qs_stride = 0;
        ks_ptr = p.k_scale + (long)p.cu_ks[b] * p.Hkv + hk;           // [sum nblk, Hkv]
        ks_tstride = p.Hkv;
    } else {
        q_off = (long)b * p.q_sb + (long)h * p.q_sh;
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

    const int row0 = qblk * BLKQ + wave * 32;        // first query row of this wave
    const int my_row = row0 + n;
    const int ntk_all = (Lk + BLKK - 1) / BLKK;      // 64-key images that exist
    int n_iters = (Lk + KT - 1) / KT;
    if (CAUSAL) {
        const int lim = (qblk * BLKQ + BLKQ + KT - 1) / KT;
        n_iters = lim < n_iters ? lim : n_iters;
    }

    // ---- Q fragments (B operand of S^T = K Q^T), resident in VGPRs ---------------------------
    v4i qf[C::KSTEPS];
    {
        const int8_t *qrow = p.q + q_off + (long)my_row * p.q_sl;
        const bool ok = my_row < Lq;
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ks++) {
            v4i z = {0, 0, 0, 0};
            qf[ks] = ok ? *reinterpret_cast<const v4i *>(qrow + 32 * ks + 16 * g) : z;
        }
    }
    // this lane's query-row scale (per-block / per-warp / per-thread granularity)
    float qsc;
    {
        int slot;
        const int rin = wave * 32 + n;               // row inside the 128-row block
        if (p.q_gran == QG_PER_BLOCK) slot = 0;
        else if (p.q_gran == QG_PER_WARP32) slot = rin >> 5;
        else if (p.q_gran == QG_PER_WARP16) slot = rin >> 4;
        else slot = (rin >> 5) * 8 + (rin & 7);      // per-thread: quant_per_thread.py:27-37
        qsc = qs_ptr[slot * qs_stride] * p.sm_scale_log2;
    }

    // ---- tile staging ------------------------------------------------------------------------
    const unsigned char *kbase = reinterpret_cast<const unsigned char *>(p.k) + k_off;
    const unsigned char *vbase = reinterpret_cast<const unsigned char *>(p.v);
    constexpr int CPR = D / 16;                                   // 16-B chunks per K row
#if SAGE_GLDS
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
                const int e = pc * 64 + lane;
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
            const unsigned char *vt = vbase + (v_tile0 + (long)tv * v_tstride) * (long)C::V_IMG_BYTES;
#pragma unroll
            for (int i = 0; i < VP / 4; i++) {
                const int pc = wave * (VP / 4) + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + pc * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(vs + hh * C::V_IMG_BYTES + pc * 1024), 16, 0, 0);
            }
        }
    };
    auto write_lds = [&](int) {};
#else
    constexpr int K_LD = C::K_TILE_BYTES / 4096, V_LD = C::V_IMG_BYTES / 4096;
    v4u kreg[K_LD], vreg[NH][V_LD];
    auto issue_loads = [&](int it, int) {
#pragma unroll
        for (int i = 0; i < K_LD; i++) {
            const int piece = tid * K_LD + i;
            const int row = piece / CPR, ch = piece % CPR;
            int key = it * KT + row;
            key = key < Lk ? key : Lk - 1;
            kreg[i] = *reinterpret_cast<const v4u *>(kbase + (long)key * p.k_sl + ch * 16);
        }
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            int tv = it * NH + hh;
            tv = tv < ntk_all ? tv : ntk_all - 1;
            const unsigned char *vt = vbase + (v_tile0 + (long)tv * v_tstride) * (long)C::V_IMG_BYTES;
#pragma unroll
            for (int i = 0; i < V_LD; i++) vreg[hh][i] = *reinterpret_cast<const v4u *>(vt + (i * 256 + tid) * 16);
        }
    };
    auto write_lds = [&](int buf) {
        unsigned char *ks = smem + buf * C::STAGE_BYTES;
        unsigned char *vs = ks + C::K_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < K_LD; i++) {
            const int piece = tid * K_LD + i;
            const int row = piece / CPR, ch = piece % CPR;
            *reinterpret_cast<v4u *>(ks + row * D + swz_chunk<D>(row, ch) * 16) = kreg[i];
        }
#pragma unroll
        for (int hh = 0; hh < NH; hh++)
#pragma unroll
            for (int i = 0; i < V_LD; i++)
                *reinterpret_cast<v4u *>(vs + hh * C::V_IMG_BYTES + (i * 256 + tid) * 16) = vreg[hh][i];
    };
#endif

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
            const long tb = (long)tk * ks_tstride;
            if (KTHREAD) {      // 4 key scales per 64 keys: token%8/2 (quant_per_thread.py:75-83); lane half g uses 2g, 2g+1
                const float s0 = ks_c[tb], s1 = ks_c[tb + 1], s2 = ks_c[tb + 2], s3 = ks_c[tb + 3];
                dst[hh][0] = g ? s2 : s0;
                dst[hh][1] = g ? s3 : s1;
            } else {
                dst[hh][0] = dst[hh][1] = ks_c[tb];
            }
        }
    };
    // ---- software-pipelined main loop ---------------------------------------------------------------
    // S^T of tile it+1 is produced (8 MFMAs) while the softmax of tile it runs on the VALU, so the matrix
    // pipe has work during the VALU-bound phase and the VALU never waits for a just-issued QK^T.
    // LDS ring of 4 stages, tiles up to it+3 in flight; K(it+1) and V(it) are resident in iteration it.
    constexpr int NSTAGE = 4;
    constexpr int DMA_PER_TILE = KP / 4 + VP / 4;                 // per wave
    auto ring_wait = [&](bool younger_in_flight) {
        if (younger_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto nact_of = [&](int it) {
        const int key0 = it * KT;
        return (key0 < Lk && (!CAUSAL || key0 <= row0 + 31)) ? 1 : 0;
    };
    auto qk_tile = [&](int it, v16i (&s)[NS]) {
        const unsigned char *ks = smem + (it & 3) * C::STAGE_BYTES;
#pragma unroll
        for (int sb = 0; sb < NS; sb++) {
#pragma unroll
            for (int i = 0; i < 16; i++) s[sb][i] = 0;
            const int krow = sb * 32 + n;
#pragma unroll
            for (int kk = 0; kk < C::KSTEPS; kk++) {
                const v4i a = *reinterpret_cast<const v4i *>(ks + krow * D + swz_chunk<D>(krow, 2 * kk + g) * 16);
                s[sb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[kk], s[sb], 0, 0, 0);
            }
        }
    };
    auto consume = [&](int it, v16i (&s)[NS], auto &&qk_next) {
        const int nact = 1;
        float mk[1][16];            // (attn_mask is not offered by this variant)
        (void)mk;
        const unsigned char *vs = smem + (it & 3) * C::STAGE_BYTES + C::K_TILE_BYTES;
        const int last_key = it * KT + BLKK - 1;
        const bool full = !(CAUSAL && last_key > row0) && (last_key < Lk);
        // ---- scales: c multiplies the raw int32 score into the log2 domain ----
        float cs[NH][2];
#pragma unroll
        for (int hh = 0; hh < NH; hh++) {
            cs[hh][0] = qsc * ksc[hh][0];
            cs[hh][1] = qsc * ksc[hh][1];
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
                mxc = fmaxf(mxc, (float)mx0 * cs[hh][0]);
                if (KTHREAD) mxc = fmaxf(mxc, (float)mx1 * cs[hh][1]);
            }
            m_new = fmaxf(m_run, pair_max(mxc) - OFF);
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int sb = 0; sb < NS; sb++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (sb < 2 * nact) {
                        const float cc = cs[sb >> 1][(KTHREAD && (i & 2)) ? 1 : 0];
                        const int key = it * KT + sb * 32 + crow(i, g);
                        const bool ok = (key < Lk) && (!CAUSAL || key <= my_row);
                        if constexpr (MASK != 0) mx = fmaxf(mx, (ok ? (float)s[sb][i] * cc : 0.0f) + mk[sb][i]);
                        else mx = fmaxf(mx, ok ? (float)s[sb][i] * cc : -INFINITY);
                    }
                }
            m_new = fmaxf(m_run, pair_max(mx) - OFF);
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
        float rs = 0.0f;
        auto p_chunk = [&](auto masked, int hh, int c, float (&e)[8]) {
            const int sb = 2 * hh + (c >> 1), r0 = (c & 1) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = r0 + j;
                const float cc = cs[hh][(KTHREAD && (i & 2)) ? 1 : 0];
                float v;
                if constexpr (MASK != 0) {
                    const int key = it * KT + sb * 32 + crow(i, g);
                    v = __builtin_amdgcn_exp2f(((key < Lk) ? (float)s[sb][i] * cc : 0.0f) + mk[sb][i] - m_new);
                } else {
                    v = __builtin_amdgcn_exp2f(__builtin_fmaf((float)s[sb][i], cc, -m_new));
                    if constexpr (decltype(masked)::value) {
                        const int key = it * KT + sb * 32 + crow(i, g);
                        const bool ok = (sb < 2 * nact) && (key < Lk) && (!CAUSAL || key <= my_row);
                        v = ok ? v : 0.0f;
                    }
                }
                e[j] = v;
                rs += v;
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
                        p_chunk(masked, hh, c, e);
                        int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], __float_as_int(e[0]), false);   // high half is overwritten next
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w0, true);
                        int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[4], e[5], __float_as_int(e[4]), false);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(e[6], e[7], w1, true);
                        pw[hh][2 * c] = w0;
                        pw[hh][2 * c + 1] = w1;
                    }
            };
            // the next tile's QK^T MFMAs go into the same basic block as this tile's exp2/pack VALU
            // work, so the scheduler can interleave them (an in-order wave cannot overlap them otherwise)
            if (full) { qk_next(); build_p(std::false_type{}); }
            else { qk_next(); build_p(std::true_type{}); }
            l_run = l_run * alpha + rs;      // lane-partial; the pair is summed in the epilogue
#pragma unroll
            for (int dt = 0; dt < C::DT; dt++) {
                const int drow = dt * 32 + n;
                v16f acc;
                if (TWO_LEVEL) {
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
                } else acc = o[dt];
#pragma unroll
                for (int hh = 0; hh < NH; hh++) {
                    if (hh < nact) {
                        const unsigned char *vr = vs + hh * C::V_IMG_BYTES + drow * 64;
                        const v4u va = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g) * 16);
                        const v4u vb = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g + 1) * 16);
#if SAGE_MXPV
                        // one K=64 block-scaled MFMA (fp8 x fp8, E8M0 scales = 127 -> x1.0)
                        const v8i av = {(int)va[0], (int)va[1], (int)va[2], (int)va[3], (int)vb[0], (int)vb[1], (int)vb[2], (int)vb[3]};
                        const v8i bv = {pw[hh][0], pw[hh][1], pw[hh][2], pw[hh][3], pw[hh][4], pw[hh][5], pw[hh][6], pw[hh][7]};
                        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#else
#define SAGE_L(lo, hi) ((long)(((unsigned long)(unsigned)(hi) << 32) | (unsigned long)(unsigned)(lo)))
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(SAGE_L(va[0], va[1]), SAGE_L(pw[hh][0], pw[hh][1]), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(SAGE_L(va[2], va[3]), SAGE_L(pw[hh][2], pw[hh][3]), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(SAGE_L(vb[0], vb[1]), SAGE_L(pw[hh][4], pw[hh][5]), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(SAGE_L(vb[2], vb[3]), SAGE_L(pw[hh][6], pw[hh][7]), acc, 0, 0, 0);
#undef SAGE_L
#endif
                    }
                }
                if (TWO_LEVEL) {
#pragma unroll
                    for (int i = 0; i < 16; i++) o[dt][i] = __builtin_fmaf(o[dt][i], alpha, acc[i]);
                } else o[dt] = acc;
            }
        } else {
            v8h pb[NH][4];
            auto build_p = [&](auto masked) {
#pragma unroll
                for (int hh = 0; hh < NH; hh++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        float e[8];
                        p_chunk(masked, hh, c, e);
#pragma unroll
                        for (int j = 0; j < 8; j++) pb[hh][c][j] = (_Float16)e[j];
                    }
            };
            // the next tile's QK^T MFMAs go into the same basic block as this tile's exp2/pack VALU
            // work, so the scheduler can interleave them (an in-order wave cannot overlap them otherwise)
            if (full) { qk_next(); build_p(std::false_type{}); }
            else { qk_next(); build_p(std::true_type{}); }
            l_run = l_run * alpha + rs;
#pragma unroll
            for (int dt = 0; dt < C::DT; dt++) {
                const int drow = dt * 32 + n;
                v16f acc;
                if (TWO_LEVEL) {
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
                } else acc = o[dt];
#pragma unroll
                for (int hh = 0; hh < NH; hh++) {
                    if (hh < nact) {
                        const unsigned char *vr = vs + hh * C::V_IMG_BYTES + drow * 128;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const v8h a = *reinterpret_cast<const v8h *>(vr + swz_chunk<128>(drow, 4 * g + c) * 16);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[hh][c], acc, 0, 0, 0);
                        }
                    }
                }
                if (TWO_LEVEL) {
#pragma unroll
                    for (int i = 0; i < 16; i++) o[dt][i] = __builtin_fmaf(o[dt][i], alpha, acc[i]);
                } else o[dt] = acc;
            }
        }
    };

    if (n_iters > 0) {
        load_kscales(0, ksc);
        issue_loads(0, 0);
    }
    if (n_iters > 1) issue_loads(1, 1);
    if (n_iters > 2) issue_loads(2, 2);
    ring_wait(n_iters > 2);                       // tiles 0 and 1 landed

    v16i sA[NS], sB[NS];
    if (n_iters > 0) qk_tile(0, sA);
    auto step = [&](auto has_next, int it, v16i (&s_use)[NS], v16i (&s_make)[NS]) {
        float ksc_next[NH][2];
        if constexpr (decltype(has_next)::value) load_kscales(it + 1, ksc_next);
        if ((it + 3) < n_iters) issue_loads(it + 3, (it + 3) & 3);
        // a wave for which the whole tile is causally masked still runs the (masked) softmax: every
        // probability is exactly 0, alpha is exactly 1 -- cheaper than breaking the basic block
        consume(it, s_use, [&]() { if constexpr (decltype(has_next)::value) qk_tile(it + 1, s_make); });
        if constexpr (decltype(has_next)::value) {
#pragma unroll
            for (int hh = 0; hh < NH; hh++) { ksc[hh][0] = ksc_next[hh][0]; ksc[hh][1] = ksc_next[hh][1]; }
        }
        ring_wait((it + 3) < n_iters);            // tile it+2 landed
    };
    {
        int it = 0;
#pragma nounroll
        for (; it + 2 < n_iters; it += 2) {
            step(std::true_type{}, it, sA, sB);
            step(std::true_type{}, it + 1, sB, sA);
        }
        if (it + 1 < n_iters) {
            step(std::true_type{}, it, sA, sB);
            step(std::false_type{}, it + 1, sB, sA);
        } else if (it < n_iters) {
            step(std::false_type{}, it, sA, sB);
        }
    }
    __syncthreads();

    // ---- epilogue: normalise, (x v_scale, + v_mean), cast, transpose through LDS, store rows ----
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
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++) {
        v4f sc4[4], mn4[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const v4f one = {1.0f, 1.0f, 1.0f, 1.0f};
            sc4[r4] = PV_FP8 ? *reinterpret_cast<const v4f *>(vsc + dt * 32 + 8 * r4 + 4 * g) : one;
        }
        if (vmn != nullptr) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) mn4[r4] = *reinterpret_cast<const v4f *>(vmn + dt * 32 + 8 * r4 + 4 * g);
        } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) { const v4f z = {0.0f, 0.0f, 0.0f, 0.0f}; mn4[r4] = z; }
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int d0 = dt * 32 + 8 * r4 + 4 * g;           // 4 consecutive d: regs 4*r4 .. 4*r4+3
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[j] = o[dt][4 * r4 + j] * inv;
                if (PV_FP8) x[j] *= sc4[r4][j];
                x[j] += mn4[r4][j];
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
    __syncthreads();
    {
        constexpr int LPR = D * 2 / 16;          // lanes per row (16 B each)
        constexpr int RPP = 64 / LPR;            // rows per pass
        unsigned char *obase = reinterpret_cast<unsigned char *>(p.o) + 2 * o_off;
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; pass++) {
            const int r = pass * RPP + lane / LPR, Q = lane % LPR;
            const v4u val = *reinterpret_cast<const v4u *>(obuf + r * (D * 2) + (Q ^ (r & 7)) * 16);
            const int grow = row0 + r;
            if (grow < Lq) *reinterpret_cast<v4u *>(obase + 2 * ((long)grow * p.o_sl) + Q * 16) = val;
        }
    }
}


template <int D, bool CAUSAL, bool KTHREAD, bool TWO_LEVEL>
static hipError_t launch_pipe_one(const AttnParams &p, int nwork, hipStream_t stream)
{
    hipLaunchKernelGGL((sage_attn_pipe_kernel<D, CAUSAL, KTHREAD, TWO_LEVEL>), dim3(nwork), dim3(256), PipeCfg<D>::LDS_BYTES, stream, p);
    return hipGetLastError();
}

hipError_t launch_attn_pipe(const AttnParams &p, int head_dim, bool causal, bool kthread, bool two_level, hipStream_t s)
{
    const int nwork = p.B * p.Hq * p.nqblk;
    if (nwork <= 0) return hipSuccess;
#define SAGE_CASE(D_, C_, K_, T_) if (head_dim == D_ && causal == C_ && kthread == K_ && two_level == T_) return launch_pipe_one<D_, C_, K_, T_>(p, nwork, s);
    SAGE_CASE(128, false, false, false) SAGE_CASE(128, false, false, true) SAGE_CASE(128, true, false, false) SAGE_CASE(128, true, false, true)
    SAGE_CASE(128, false, true, false)  SAGE_CASE(128, false, true, true)  SAGE_CASE(128, true, true, false)  SAGE_CASE(128, true, true, true)
    SAGE_CASE(64, false, false, false)  SAGE_CASE(64, false, false, true)  SAGE_CASE(64, true, false, false)  SAGE_CASE(64, true, false, true)
    SAGE_CASE(64, false, true, false)   SAGE_CASE(64, false, true, true)   SAGE_CASE(64, true, true, false)   SAGE_CASE(64, true, true, true)
#undef SAGE_CASE
    return hipErrorInvalidValue;
}

}  // namespace sage
