// sage_attn_loop_f8.h -- the software-pipelined key loop of sage_attn_kernel for FP8 PV: a code fragment, included INSIDE the kernel body
// (sage_attn_kernel.h, `if constexpr (PV_FP8)`), where every name it uses is in scope (o, qf, smem, it, n_steady, diag_ok, tail_ok, issue_loads, ...).
// Not a header of its own: no include guard, no declarations.
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
                    SAGE_TSTAMP(8);                  // (trace builds: first scores formed / six-body trips done / remainder done / last bodies done)
                    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
#pragma nounroll
                    for (; n6 > 0; n6--) {
                        body(I0{}, I0{}, n, g, sA, sB, pA, pB); body(I1{}, I0{}, n, g, sB, sA, pB, pA); body(I2{}, I0{}, n, g, sA, sB, pA, pB);
                        body(I0{}, I0{}, n, g, sB, sA, pB, pA); body(I1{}, I0{}, n, g, sA, sB, pA, pB); body(I2{}, I0{}, n, g, sB, sA, pB, pA);
                    }
                    // (the remainder body's per-lane LDS offsets are derived behind the six-body loop from a lane index the compiler cannot see
                    //  through: formed in front of it they would stay live across it, next to that loop's own -- registers D = 64 does not have)
                    SAGE_TSTAMP(13);
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
                    SAGE_TSTAMP(14);
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
                    SAGE_TSTAMP(15);
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
