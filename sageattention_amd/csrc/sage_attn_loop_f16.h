// sage_attn_loop_f16.h -- the software-pipelined key loop of sage_attn_kernel for FP16 PV: a code fragment, included INSIDE the kernel body
// (sage_attn_kernel.h, the `else` of `if constexpr (PV_FP8)`), where every name it uses is in scope.  Not a header of its own.
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
