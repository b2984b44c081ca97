// sage_attn.hip -- fused INT8-QK^T / online-softmax / FP8-or-FP16-PV attention for gfx950.
//
// Replaces (behaviourally, not textually) the reference kernels
//   csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:46-704   (INT8 QK, FP8 PV, two-level accumulation)
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
//    O^T = V^T P^T (v_mfma_f32_32x32x16_fp8_fp8 / _f16): the K/V pre-pass stores V^T tiles in
//    the matching "position" order (sage_common.h), so no LDS round trip for P.
//  * two-level accumulation: every 64-key tile's product starts from a zero accumulator and is
//    folded into the FP32 running output with one FMA (O = O*alpha + T).
//  * K/V tiles are double-buffered in LDS with XOR-swizzled 16-byte chunks (conflict-free
//    ds_read_b128); next tile's global loads are issued before the current tile's MFMAs and
//    written to LDS after them (register-staged, one barrier per tile).
//  * output tile is transposed through (now free) LDS and stored as whole rows, 16 B per lane.
#include "sage_common.h"
#include "sage_kernels.h"

// ---- build-time variant switches (A/B-tested on the GPU; see DESIGN.md "kernel ladder") ----------
#ifndef SAGE_MAGIC      // seed the int32 QK^T accumulator with 0x4B400000 so the result bits ARE the float
#define SAGE_MAGIC 1    // 12582912 + dot: no v_cvt_f32_i32, the offset folds into the exp2 FMA addend
#endif
#ifndef SAGE_GLDS       // K/V tiles by LDS-DMA (global_load_lds_dwordx4) instead of VGPR staging
#define SAGE_GLDS 1
#endif
#ifndef SAGE_MXPV       // FP8 PV on v_mfma_scale_f32_32x32x64_f8f6f4 with unit E8M0 scales (2x rate)
#define SAGE_MXPV 1
#endif
#ifndef SAGE_SETPRIO    // s_setprio 1 around MFMA clusters
#define SAGE_SETPRIO 0
#endif
#ifndef SAGE_ABL        // timing-only ablations (WRONG results): 1 no exp, 2 no PV MFMA, 4 no QK MFMA,
#define SAGE_ABL 0      // 8 no O update, 16 no max/sum, 32 no barrier
#endif

namespace sage {

constexpr int kMagicI = 0x4B400000;          // bits of 12582912.0f = 2^23 + 2^22: ulp 1 over +-2^22
constexpr float kMagicF = 12582912.0f;

template <int D, bool PV_FP8> struct TileCfg {
    static constexpr int K_ROW_BYTES = D;                       // int8
    static constexpr int K_TILE_BYTES = BLKK * D;
    static constexpr int V_ROW_BYTES = PV_FP8 ? 64 : 128;       // 64 positions per d row
    static constexpr int V_TILE_BYTES = D * V_ROW_BYTES;
    static constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;
    static constexpr int O_BYTES = BLKQ * D * 2;
    static constexpr int LDS_BYTES = (2 * STAGE_BYTES > O_BYTES) ? 2 * STAGE_BYTES : O_BYTES;
    static constexpr int KSTEPS = D / 32;                       // i8 MFMA k-steps over head dim
    static constexpr int DT = D / 32;                           // 32-wide output d tiles
    static constexpr int K_LD = K_TILE_BYTES / (256 * 16);      // 16-B pieces per thread
    static constexpr int V_LD = V_TILE_BYTES / (256 * 16);
};

// c/d register r of a 32x32 MFMA tile -> row index inside the tile (lane half g = lane>>5)
__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

template <int D, bool PV_FP8, bool CAUSAL, bool KTHREAD, bool TWO_LEVEL>
__global__ void __launch_bounds__(256, 2)
sage_attn_kernel(const AttnParams p)
{
    using C = TileCfg<D, PV_FP8>;
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
    long v_tile0, v_tstride;              // V tile index = v_tile0 + t * v_tstride
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
    const int ntk_all = (Lk + BLKK - 1) / BLKK;
    int n_tiles = ntk_all;
    if (CAUSAL) {
        const int lim = (qblk * BLKQ + BLKQ + BLKK - 1) / BLKK;
        n_tiles = lim < n_tiles ? lim : n_tiles;
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

    // ---- tile staging: global -> VGPR -> LDS -------------------------------------------------
    const unsigned char *kbase = reinterpret_cast<const unsigned char *>(p.k) + k_off;
    const unsigned char *vbase = reinterpret_cast<const unsigned char *>(p.v);
#if SAGE_GLDS
    // LDS-DMA: every wave-instruction moves 64 x 16 B = 1 KiB; the LDS destination is lane-linear
    // (M0 base + lane*16), so the XOR swizzle of the K image goes on the per-lane SOURCE address.
    // Key rows past Lk are clamped to the last valid row (their scores are masked anyway).
    auto issue_loads = [&](int t, int buf) {
        unsigned char *ks = smem + buf * C::STAGE_BYTES;
        unsigned char *vs = ks + C::K_TILE_BYTES;
        constexpr int CPR = D / 16;
        constexpr int KP = C::K_TILE_BYTES / 1024, VP = C::V_TILE_BYTES / 1024;   // 1-KiB pieces
#pragma unroll
        for (int i = 0; i < KP / 4; i++) {
            const int pc = wave * (KP / 4) + i;
            const int e = pc * 64 + lane;                      // 16-B slot index inside the tile
            const int row = e / CPR, phys = e % CPR;
            int key = t * BLKK + row;
            key = key < Lk ? key : Lk - 1;
            const unsigned char *src = kbase + (long)key * p.k_sl + swz_chunk<D>(row, phys) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(ks + pc * 1024), 16, 0, 0);
        }
        const unsigned char *vt = vbase + (v_tile0 + (long)t * v_tstride) * (long)C::V_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < VP / 4; i++) {
            const int pc = wave * (VP / 4) + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vt + pc * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(vs + pc * 1024), 16, 0, 0);
        }
    };
    auto write_lds = [&](int) {};
#else
    v4u kreg[C::K_LD], vreg[C::V_LD];

    auto issue_loads = [&](int t, int) {
        // K tile: 64 rows x D bytes; thread -> (row, 16-B chunk), 16B*K_LD contiguous per thread
        constexpr int CPR = D / 16;                              // chunks per row
#pragma unroll
        for (int i = 0; i < C::K_LD; i++) {
            const int piece = tid * C::K_LD + i;
            const int row = piece / CPR, ch = piece % CPR;
            const int key = t * BLKK + row;
            v4u z = {0u, 0u, 0u, 0u};
            kreg[i] = (key < Lk) ? *reinterpret_cast<const v4u *>(kbase + (long)key * p.k_sl + ch * 16) : z;
        }
        const unsigned char *vt = vbase + (v_tile0 + (long)t * v_tstride) * (long)C::V_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < C::V_LD; i++)
            vreg[i] = *reinterpret_cast<const v4u *>(vt + (i * 256 + tid) * 16);
    };
    auto write_lds = [&](int buf) {
        unsigned char *ks = smem + buf * C::STAGE_BYTES;
        unsigned char *vs = ks + C::K_TILE_BYTES;
        constexpr int CPR = D / 16;
#pragma unroll
        for (int i = 0; i < C::K_LD; i++) {
            const int piece = tid * C::K_LD + i;
            const int row = piece / CPR, ch = piece % CPR;
            *reinterpret_cast<v4u *>(ks + row * D + swz_chunk<D>(row, ch) * 16) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < C::V_LD; i++)
            *reinterpret_cast<v4u *>(vs + (i * 256 + tid) * 16) = vreg[i];   // image is pre-swizzled
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

#if SAGE_MAGIC
    v16i magic;
#pragma unroll
    for (int i = 0; i < 16; i++) magic[i] = kMagicI;
#endif
    if (n_tiles > 0) {
        issue_loads(0, 0);
        write_lds(0);
    }
    __syncthreads();

    for (int t = 0; t < n_tiles; t++) {
        const int cur = t & 1;
        const bool more = (t + 1) < n_tiles;
        if (more) issue_loads(t + 1, cur ^ 1);

        // wave-uniform: does this wave have any unmasked key in the tile?
        const bool active = !CAUSAL || (t * BLKK <= row0 + 31);
        if (active) {
            const unsigned char *ks = smem + cur * C::STAGE_BYTES;
            const unsigned char *vs = ks + C::K_TILE_BYTES;

            // ---- S^T = K Q^T (int8 -> int32) ----
            v16i s[2];
#if SAGE_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const int krow = sub * 32 + n;
#pragma unroll
                for (int kk = 0; kk < C::KSTEPS; kk++) {
                    const v4i a = *reinterpret_cast<const v4i *>(ks + krow * D + swz_chunk<D>(krow, 2 * kk + g) * 16);
#if SAGE_ABL & 4
                    if (kk == 0) {
#pragma unroll
                        for (int i = 0; i < 16; i++) s[sub][i] = qf[0][i & 3] + a[i & 3];
                    }
#elif SAGE_MAGIC
                    s[sub] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[kk], kk == 0 ? magic : s[sub], 0, 0, 0);
#else
                    v16i z;
#pragma unroll
                    for (int i = 0; i < 16; i++) z[i] = 0;
                    s[sub] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[kk], kk == 0 ? z : s[sub], 0, 0, 0);
#endif
                }
            }
#if SAGE_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            // score register -> float.  With SAGE_MAGIC the accumulator bits already are the float
            // (kMagicF + dot), exact for |dot| < 2^22 (max |dot| = 128*127*127 < 2^21).
#if SAGE_MAGIC
#define SAGE_SF(sub, i) __int_as_float(s[sub][i])
            constexpr float SHIFT = kMagicF;
#else
#define SAGE_SF(sub, i) ((float)s[sub][i])
            constexpr float SHIFT = 0.0f;
#endif

            // ---- scales: c[sel] multiplies the raw int32 score into the log2 domain ----
            float c0, c1;
            if (KTHREAD) {      // 4 key scales per 64-key tile: token%8/2 (quant_per_thread.py:75-83)
                c0 = qsc * ks_ptr[(long)t * ks_tstride + 2 * g];
                c1 = qsc * ks_ptr[(long)t * ks_tstride + 2 * g + 1];
            } else {
                c0 = c1 = qsc * ks_ptr[(long)t * ks_tstride];
            }
            const float sh0 = SHIFT * c0, sh1 = SHIFT * c1;   // shift of the magic seed in the log2 domain

            // ---- online softmax ----
            const bool need_mask = (CAUSAL && (t * BLKK + BLKK - 1 > row0)) || (t * BLKK + BLKK > Lk);
            float pf[2][16];
            float m_new;
            if (!need_mask) {
                float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
#if !(SAGE_ABL & 16)
                        if (KTHREAD && (i & 2)) mx1 = fmaxf(mx1, SAGE_SF(sub, i));
                        else mx0 = fmaxf(mx0, SAGE_SF(sub, i));
#else
                        if (i == 0) { mx0 = SAGE_SF(sub, 0); mx1 = SAGE_SF(sub, 2); }
#endif
                    }
                float mx = __builtin_fmaf(mx0, c0, -sh0);
                if (KTHREAD) mx = fmaxf(mx, __builtin_fmaf(mx1, c1, -sh1));
                mx = pair_max(mx);
                m_new = fmaxf(m_run, mx - OFF);
                const float a0 = -(m_new + sh0), a1 = -(m_new + sh1);
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const bool hi = KTHREAD && (i & 2);
#if SAGE_ABL & 1
                        pf[sub][i] = __builtin_fmaf(SAGE_SF(sub, i), hi ? c1 : c0, hi ? a1 : a0);
#else
                        pf[sub][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(SAGE_SF(sub, i), hi ? c1 : c0, hi ? a1 : a0));
#endif
                    }
            } else {
                float mx = -INFINITY;
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const bool hi = KTHREAD && (i & 2);
                        const int key = t * BLKK + sub * 32 + crow(i, g);
                        const bool ok = (key < Lk) && (!CAUSAL || key <= my_row);
                        const float v = ok ? __builtin_fmaf(SAGE_SF(sub, i), hi ? c1 : c0, hi ? -sh1 : -sh0) : -INFINITY;
                        pf[sub][i] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = pair_max(mx);
                m_new = fmaxf(m_run, mx - OFF);
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        pf[sub][i] = __builtin_amdgcn_exp2f(pf[sub][i] - m_new);
            }
#undef SAGE_SF
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float rs = 0.0f;
#if !(SAGE_ABL & 16)
#pragma unroll
            for (int sub = 0; sub < 2; sub++)
#pragma unroll
                for (int i = 0; i < 16; i++) rs += pf[sub][i];
#else
            rs = pf[0][0] + pf[1][5];
#endif
            l_run = l_run * alpha + rs;          // lane-partial; the pair is summed in the epilogue

            // ---- P -> low precision, already in PV B-operand order ----
            //      chunk c = 2*sub + u takes registers 8u..8u+7 of S^T tile `sub`
            if (!TWO_LEVEL) {
#pragma unroll
                for (int dt = 0; dt < C::DT; dt++)
#pragma unroll
                    for (int i = 0; i < 16; i++) o[dt][i] *= alpha;
            }
            if constexpr (PV_FP8) {
                long pb[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int sub = c >> 1, r0 = (c & 1) * 8;
                    int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(pf[sub][r0 + 0], pf[sub][r0 + 1], 0, false);
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(pf[sub][r0 + 2], pf[sub][r0 + 3], w0, true);
                    int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(pf[sub][r0 + 4], pf[sub][r0 + 5], 0, false);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(pf[sub][r0 + 6], pf[sub][r0 + 7], w1, true);
                    pb[c] = (long)(((unsigned long)(unsigned)w1 << 32) | (unsigned long)(unsigned)w0);
                }
#pragma unroll
                for (int dt = 0; dt < C::DT; dt++) {
                    const int drow = dt * 32 + n;
                    const unsigned char *vr = vs + drow * 64;
                    const v4u va = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g) * 16);
                    const v4u vb = *reinterpret_cast<const v4u *>(vr + swz_chunk<64>(drow, 2 * g + 1) * 16);
                    v16f acc;
                    if (TWO_LEVEL) {
#pragma unroll
                        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
                    } else acc = o[dt];
#if SAGE_SETPRIO
                    __builtin_amdgcn_s_setprio(1);
#endif
#if SAGE_MXPV
                    // one K=64 block-scaled MFMA (fp8 x fp8, E8M0 scales = 127 -> x1.0) per d tile:
                    // same products, same FP32 accumulation, twice the rate of 4 x 32x32x16
                    const v8i av = {(int)va[0], (int)va[1], (int)va[2], (int)va[3], (int)vb[0], (int)vb[1], (int)vb[2], (int)vb[3]};
                    const v8i bv = {(int)pb[0], (int)(pb[0] >> 32), (int)pb[1], (int)(pb[1] >> 32),
                                    (int)pb[2], (int)(pb[2] >> 32), (int)pb[3], (int)(pb[3] >> 32)};
#if SAGE_ABL & 2
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] += __int_as_float(av[i & 7] ^ bv[i & 7]);
#else
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#endif
#else
                    const long a0 = (long)(((unsigned long)va[1] << 32) | va[0]);
                    const long a1 = (long)(((unsigned long)va[3] << 32) | va[2]);
                    const long a2 = (long)(((unsigned long)vb[1] << 32) | vb[0]);
                    const long a3 = (long)(((unsigned long)vb[3] << 32) | vb[2]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, pb[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, pb[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2, pb[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a3, pb[3], acc, 0, 0, 0);
#endif
#if SAGE_SETPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
#if SAGE_ABL & 8
                    o[dt][dt] += acc[0] + acc[5] + acc[10] + acc[15];
#else
                    if (TWO_LEVEL) {
#pragma unroll
                        for (int i = 0; i < 16; i++) o[dt][i] = __builtin_fmaf(o[dt][i], alpha, acc[i]);
                    } else o[dt] = acc;
#endif
                }
            } else {
                v8h pb[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int sub = c >> 1, r0 = (c & 1) * 8;
#pragma unroll
                    for (int j = 0; j < 8; j++) pb[c][j] = (_Float16)pf[sub][r0 + j];
                }
#pragma unroll
                for (int dt = 0; dt < C::DT; dt++) {
                    const int drow = dt * 32 + n;
                    const unsigned char *vr = vs + drow * 128;
                    v16f acc;
                    if (TWO_LEVEL) {
#pragma unroll
                        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
                    } else acc = o[dt];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const v8h a = *reinterpret_cast<const v8h *>(vr + swz_chunk<128>(drow, 4 * g + c) * 16);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[c], acc, 0, 0, 0);
                    }
                    if (TWO_LEVEL) {
#pragma unroll
                        for (int i = 0; i < 16; i++) o[dt][i] = __builtin_fmaf(o[dt][i], alpha, acc[i]);
                    } else o[dt] = acc;
                }
            }
        }

        if (more) write_lds(cur ^ 1);
#if SAGE_ABL & 32
        __builtin_amdgcn_s_waitcnt(0);
#else
        __syncthreads();
#endif
    }

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
    const float *vsc = PV_FP8 ? p.v_scale + ((long)b * p.Hkv + hk) * D : nullptr;
    const float *vmn = (p.v_mean != nullptr) ? p.v_mean + ((long)b * p.Hkv + hk) * D : nullptr;
#pragma unroll
    for (int dt = 0; dt < C::DT; dt++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int d0 = dt * 32 + 8 * r4 + 4 * g;           // 4 consecutive d: regs 4*r4 .. 4*r4+3
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[j] = o[dt][4 * r4 + j] * inv;
                if (PV_FP8) x[j] *= vsc[d0 + j];
                if (vmn != nullptr) x[j] += vmn[d0 + j];
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

// ---------------------------------------------------------------------------------------------
template <int D, bool PV_FP8, bool CAUSAL, bool KTHREAD, bool TWO_LEVEL>
static hipError_t launch_one(const AttnParams &p, int nwork, hipStream_t stream)
{
    using C = TileCfg<D, PV_FP8>;
    auto kern = sage_attn_kernel<D, PV_FP8, CAUSAL, KTHREAD, TWO_LEVEL>;
#ifdef SAGE_LDS_MIN_BYTES   // experiments: cap workgroups per CU through the LDS budget
    constexpr int lds = C::LDS_BYTES > SAGE_LDS_MIN_BYTES ? C::LDS_BYTES : SAGE_LDS_MIN_BYTES;
    if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#else
    constexpr int lds = C::LDS_BYTES;
#endif
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int D, bool PV_FP8>
static hipError_t launch_d(const AttnParams &p, int nwork, bool causal, bool kthread, bool two_level, hipStream_t s)
{
#define SAGE_CASE(C_, K_, T_) if (causal == C_ && kthread == K_ && two_level == T_) return launch_one<D, PV_FP8, C_, K_, T_>(p, nwork, s);
    SAGE_CASE(false, false, false) SAGE_CASE(false, false, true)
    SAGE_CASE(true, false, false)  SAGE_CASE(true, false, true)
    SAGE_CASE(false, true, false)  SAGE_CASE(false, true, true)
    SAGE_CASE(true, true, false)   SAGE_CASE(true, true, true)
#undef SAGE_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_attn(const AttnParams &p, int head_dim, bool pv_fp8, bool causal, bool kthread,
                       bool two_level, hipStream_t stream)
{
    const int nwork = p.B * p.Hq * p.nqblk;
    if (nwork <= 0) return hipSuccess;
    if (head_dim == 128) return pv_fp8 ? launch_d<128, true>(p, nwork, causal, kthread, two_level, stream)
                                       : launch_d<128, false>(p, nwork, causal, kthread, two_level, stream);
    if (head_dim == 64) return pv_fp8 ? launch_d<64, true>(p, nwork, causal, kthread, two_level, stream)
                                      : launch_d<64, false>(p, nwork, causal, kthread, two_level, stream);
    return hipErrorInvalidValue;
}

}  // namespace sage
