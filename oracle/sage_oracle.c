/*
 * sage_oracle.c -- CPU restatement of SageAttention's quantized attention path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in sageattention_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU timing baseline.
 *
 * Parity pin: the reference ships no tests or golden vectors, so
 * this restatement is pinned against outputs of the reference's own Triton
 * kernels run on CPU under TRITON_INTERPRET=1 (tests/golden/gen_golden.py wrote
 * the .npz fixtures under tests/golden from /root/reference) and against fp32 SDPA.  The FP8-PV
 * path exists in the reference only as CUDA (not buildable here: nvcc/PTX), so
 * for that path parity is pinned to the algorithm text cited below plus fp32
 * SDPA bounds -- "parity partially pinned" (see DESIGN.md).
 *
 * Each function cites the reference file:line (relative to /root/reference) it
 * follows.  Plain C99 + optional OpenMP; build with oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* scalar format conversions (round-to-nearest-even)                         */
/* ------------------------------------------------------------------------ */

static inline uint32_t f_bits(float f) { uint32_t x; memcpy(&x, &f, 4); return x; }
static inline float bits_f(uint32_t x) { float f; memcpy(&f, &x, 4); return f; }

ORC_EXPORT float orc_h2f(uint16_t h)
{
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    if (e == 0) {
        if (m == 0) return bits_f(s);
        float v = (float)m * 5.9604644775390625e-08f; /* 2^-24 */
        return (s ? -v : v);
    }
    if (e == 31) return bits_f(s | 0x7f800000u | (m << 13));
    return bits_f(s | ((e + 112) << 23) | (m << 13));
}

ORC_EXPORT uint16_t orc_f2h(float f)
{
    uint32_t x = f_bits(f), s = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);          /* >= 65520 -> inf */
    if (x < 0x38800000u) {                                         /* |f| < 2^-14: subnormal half */
        if (x < 0x33000000u) return (uint16_t)s;                   /* < 2^-25 -> 0 */
        uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u, sh = 126 - e; /* sh in [14,24] */
        uint32_t r = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(s | r);
    }
    uint32_t e = (x >> 23) - 112, m = x & 0x7fffffu;
    uint32_t r = (e << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(s | r);
}

ORC_EXPORT float orc_bf2f(uint16_t b) { return bits_f((uint32_t)b << 16); }

ORC_EXPORT uint16_t orc_f2bf(float f)
{
    uint32_t x = f_bits(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
    uint32_t lsb = (x >> 16) & 1;
    x += 0x7fffu + lsb;
    return (uint16_t)(x >> 16);
}

/* OCP e4m3fn, RNE, saturate-to-finite (the reference's cvt.rn.satfinite.e4m3x2.f32,
 * csrc/numeric_conversion.cuh:46-61).  448 = 0x7e is the largest finite value. */
ORC_EXPORT uint8_t orc_f2e4m3(float f)
{
    uint32_t x = f_bits(f);
    uint8_t s = (uint8_t)((x >> 24) & 0x80u);
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint8_t)(s | 0x7f);
    float a = bits_f(x);
    if (a >= 448.0f) return (uint8_t)(s | 0x7e);
    if (a < 0.015625f) {                                  /* < 2^-6: subnormal, unit 2^-9 */
        int r = (int)nearbyintf(a * 512.0f);              /* default rounding mode = RNE */
        return (uint8_t)(s | (uint8_t)r);
    }
    int ex = (int)(x >> 23) - 127;                        /* in [-6, 8] */
    uint32_t m = x & 0x7fffffu;
    uint32_t r = ((uint32_t)(ex + 7) << 3) | (m >> 20), rem = m & 0xfffffu;
    if (rem > 0x80000u || (rem == 0x80000u && (r & 1))) r++;
    if (r > 0x7e) r = 0x7e;
    return (uint8_t)(s | r);
}

ORC_EXPORT float orc_e4m3_2f(uint8_t b)
{
    int s = b & 0x80, e = (b >> 3) & 0xf, m = b & 7;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;              /* 2^-9 */
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + (float)m * 0.125f, e - 7);
    return s ? -v : v;
}

static inline float ld16(const uint16_t *p, int dtype) { return dtype == 0 ? orc_h2f(*p) : orc_bf2f(*p); }
static inline uint16_t st16(float f, int dtype) { return dtype == 0 ? orc_f2h(f) : orc_f2bf(f); }

ORC_EXPORT void orc_convert_array(const float *in, void *out, long n, int kind)
{   /* kind 0: f32->f16, 1: f32->bf16, 2: f32->e4m3 (conversion self-test hook) */
    for (long i = 0; i < n; i++) {
        if (kind == 0) ((uint16_t *)out)[i] = orc_f2h(in[i]);
        else if (kind == 1) ((uint16_t *)out)[i] = orc_f2bf(in[i]);
        else ((uint8_t *)out)[i] = orc_f2e4m3(in[i]);
    }
}

/* ------------------------------------------------------------------------ */
/* INT8 quantisation of Q / K                                                */
/* ------------------------------------------------------------------------ */
/*
 * x:      [B,H,L,D] fp16 (dtype 0) or bf16 (dtype 1), contiguous
 * mean:   [B,H,D] same dtype or NULL (K smoothing mean km, core.py:279-295)
 * group:  [L] int32 -- scale-group index of every row (rows of one group share a
 *         scale); ngroups scales per (b,h): scale[B,H,ngroups].  The caller
 *         builds it per granularity:
 *           per-block   row/BLK                              quant_per_block.py:29-31
 *           per-warp    row/WARPQ                            quant.py:169-171, fused.cu:685-768
 *           per-thread  Q: (row/WARPQ)*8 + row%8             quant_per_thread.py:27-37
 *                       K: (row/WARPK)*4 + (row%8)/2         quant_per_thread.py:75-83
 * style 0 ("triton", quant_per_block.py:39-47): x*=pre_scale; scale=amax/127;
 *         q = x/scale; q += 0.5*sign; truncate.  A K mean is subtracted in the
 *         input dtype first (`k = k - km` in torch, quant_per_block.py:53-54).
 * style 1 ("cuda", fused.cu:110-186): x = (x - mean) in fp32, x*=pre_scale,
 *         amax floor 1e-7, q = rint_sat(x * (127/amax))  (numeric_conversion.cuh:144-149)
 * style 2 ("triton per-thread", quant_per_thread.py:41-44): as style 0 but
 *         scale = amax/127 + 1e-7.
 * All-zero group in style 0: the reference computes 0/0 (NaN, undefined int
 * cast); this oracle defines scale=0, q=0 for that case.
 */
ORC_EXPORT int orc_quant_int8(const uint16_t *x, int dtype, const uint16_t *mean, int8_t *out,
                              float *scale, const int32_t *group, int ngroups,
                              int B, int H, int L, int D, float pre_scale, int style)
{
    float *amax = (float *)malloc(sizeof(float) * (size_t)ngroups);
    float *row = (float *)malloc(sizeof(float) * (size_t)L * D);
    if (!amax || !row) return -1;
    for (int b = 0; b < B; b++)
        for (int h = 0; h < H; h++) {
            const uint16_t *xb = x + ((size_t)(b * H + h) * L) * D;
            const uint16_t *mb = mean ? mean + (size_t)(b * H + h) * D : NULL;
            int8_t *ob = out + ((size_t)(b * H + h) * L) * D;
            float *sb = scale + (size_t)(b * H + h) * ngroups;
            for (int g = 0; g < ngroups; g++) amax[g] = (style == 1) ? 1e-7f : 0.0f;
            for (int l = 0; l < L; l++)
                for (int d = 0; d < D; d++) {
                    float v = ld16(xb + (size_t)l * D + d, dtype);
                    if (mb) {
                        v = v - ld16(mb + d, dtype);
                        if (style != 1) v = ld16(&(uint16_t){st16(v, dtype)}, dtype);
                    }
                    v *= pre_scale;
                    row[(size_t)l * D + d] = v;
                    float a = fabsf(v);
                    if (a > amax[group[l]]) amax[group[l]] = a;
                }
            for (int g = 0; g < ngroups; g++) {
                float sc = amax[g] / 127.0f;
                if (style == 2) sc += 1e-7f;
                sb[g] = sc;
            }
            for (int l = 0; l < L; l++) {
                int g = group[l];
                for (int d = 0; d < D; d++) {
                    float v = row[(size_t)l * D + d];
                    int q;
                    if (style == 1) {
                        float t = v * (127.0f / amax[g]);
                        t = nearbyintf(t);
                        if (t > 127.0f) t = 127.0f;
                        if (t < -128.0f) t = -128.0f;
                        q = (int)t;
                    } else {
                        float sc = sb[g];
                        if (sc == 0.0f) q = 0;
                        else {
                            float t = v / sc;
                            t += (t >= 0.0f) ? 0.5f : -0.5f;
                            q = (int)t; /* truncation toward zero */
                            if (q > 127) q = 127;
                            if (q < -128) q = -128;
                        }
                    }
                    ob[(size_t)l * D + d] = (int8_t)q;
                }
            }
        }
    free(amax);
    free(row);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* FP8 per-channel quantisation of V                                         */
/* ------------------------------------------------------------------------ */
/*
 * Follows quant.py:224-293 + MeanScaleKernel fused.cu:316-427 (smooth_v=False
 * branch): per (b,h,d) amax over tokens, scale = amax/scale_max,
 * v8 = e4m3_rne_sat(v * (scale_max/amax)).  Output is kept in the LOGICAL
 * layout [B,H,L,D] (one byte per element): the reference's [D, L] transpose and
 * its 16-token permutation (fused.cu:287-291) are NVIDIA operand-layout details,
 * just as the gfx950 tiled layout is private to the HIP kernels.
 * amax==0 gives scale 0 and zeros here (reference: inf/NaN).
 */
ORC_EXPORT int orc_quant_v_fp8(const uint16_t *v, int dtype, uint8_t *out, float *v_scale, const float *mean,
                               int B, int H, int L, int D, float scale_max)
{
    /* mean (nullable, [B,H,D]): smooth_v -- subtracted before scaling; amax becomes
     * max(|max - mean|, |min - mean|) (fused.cu:383-385).  The reference computes the mean itself as
     * sum / ceil16(L) (fused.cu:335,381); it is an input here so that the checker and the checked
     * kernel quantise against the very same mean.
     * MeanScaleKernel reads ceil16(L) tokens of the ZERO-PADDED transpose (fused.cu:335-357, padding written
     * by TransposePadPermuteKernel fused.cu:283-286), so when L % 16 != 0 the zeros take part in max and min:
     * immaterial for the plain amax, but with smooth_v the amax includes |0 - mean|. */
    for (int b = 0; b < B; b++)
        for (int h = 0; h < H; h++) {
            const uint16_t *vb = v + ((size_t)(b * H + h) * L) * D;
            uint8_t *ob = out + ((size_t)(b * H + h) * L) * D;
            float *sb = v_scale + (size_t)(b * H + h) * D;
            for (int d = 0; d < D; d++) {
                const float mu = mean ? mean[(size_t)(b * H + h) * D + d] : 0.0f;
                float mx = -INFINITY, mn = INFINITY;
                for (int l = 0; l < L; l++) {
                    float f = ld16(vb + (size_t)l * D + d, dtype);
                    mx = fmaxf(mx, f);
                    mn = fminf(mn, f);
                }
                if (L % 16) { mx = fmaxf(mx, 0.0f); mn = fminf(mn, 0.0f); }
                float amax = fmaxf(fabsf(mx - mu), fabsf(mn - mu));
                sb[d] = amax / scale_max;
                float recp = amax > 0.0f ? scale_max / amax : 0.0f;
                for (int l = 0; l < L; l++) {
                    float f = ld16(vb + (size_t)l * D + d, dtype);
                    if (mean) f -= mu;
                    ob[(size_t)l * D + d] = orc_f2e4m3(f * recp);
                }
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* fused attention on quantised operands                                     */
/* ------------------------------------------------------------------------ */
/*
 * q  [B,Hq,Lq,D] int8, k [B,Hkv,Lk,D] int8 (contiguous)
 * v  pv_mode 0/1: [B,Hkv,Lk,D] fp16 bits; pv_mode 2/3: [B,Hkv,Lk,D] e4m3 bytes
 * q_scale [B,Hq,nqs], q_sidx[Lq]: scale slot of each query row
 * k_scale [B,Hkv,nks], k_sidx[Lk]: scale slot of each key
 * v_scale [B,Hkv,D] (fp8 modes) or NULL; v_mean [B,Hkv,D] or NULL: added after normalisation
 *   (smooth_v: epilogue order normalise -> x v_scale -> + v_mean, qk_int_sv_f8_cuda_sm89.cuh:572-656)
 * c: multiplier applied to the dequantised score; 1.0 when sm_scale*log2e was
 *    folded into Q (quant_per_block.py:87), sm_scale*log2e otherwise
 *    (qk_int_sv_f8_cuda_sm89.cuh:298,334-335).
 * pv_mode 0: Triton path (attn_qk_int8_per_block.py:33-66, _causal.py:33-60):
 *            P->fp16, tile product accumulated then rounded to fp16
 *            (tl.dot(..., out_dtype=fp16) as the CPU interpreter evaluates it:
 *            fp32 sum rounded once), added to the fp32 accumulator.
 * pv_mode 1: fp16 P, fp32 tile product, fp32 accumulator (sm80 "fp32",
 *            qk_int_sv_f16_cuda_sm80.cu:303-420 in spirit).  The softmax denominator is the FP32
 *            sum of the fp16-ROUNDED P: the kernels are instantiated with
 *            DenominatorAccumUnit = kTensorCore (qk_int_sv_f16_cuda_sm80.cu:814,989,1164,1348), i.e.
 *            RS_32_to_16 first, then accumulate_d -> mma::rowsum_f16f16f32 on the packed halves
 *            (qk_int_sv_f16_cuda_sm80.cu:313-320, attn_utils.cuh:529-545).  pv_mode 0 (Triton) and the
 *            FP8 modes sum the un-rounded exponentials (attn_qk_int8_per_block.py:57-60,
 *            qk_int_sv_f8_cuda_sm90.cu:317-318).
 * pv_mode 2: FP8 PV, exp offset 8.807 (attn_utils.cuh:30,377-389), P->e4m3
 *            (attn_utils.cuh:478-493), per-tile product started from zero and
 *            added to the fp32 accumulator = two-level accumulation
 *            (attn_utils.cuh:813-894), epilogue O/l*v_scale
 *            (qk_int_sv_f8_cuda_sm89.cuh:572-621).
 * pv_mode 3: as 2 but accumulating straight into O (single level).
 * Causal mask is top-left aligned: key > query masked (attn_utils.cuh:308-310).
 * Tiles: 128 query rows x 64 keys (attn_qk_int8_per_block.py:131-132).
 * lse (nullable) [B,Hq,Lq] = log2(l) + m, log2 units (attn_qk_int8_per_block.py:126-127).
 * out dtype: 0 fp16, 1 bf16.
 * score_mode (FP8 modes 2 / 3 only; 0 everywhere else):
 *   0 "exact":  P = exp2(fma(raw score, sm_scale', -m)), the reference's formula (attn_utils.cuh:445-449) -- the mode that is pinned to
 *               the reference text and that every gfx950 kernel runs by default (since round 6).
 *   1 "folded": the SAME formula reassociated as the gfx950 kernels' OPT-IN FP8 variant (SAGE_ATTR_FP8_FOLDED_SCORES) evaluates it.  It reads the INT32 accumulator,
 *               which starts from the bit pattern 0x3E22F983 (the float 1/(2 pi)), as the float  x = bias + s * 2^-26  (exact for
 *               |s| <= 2^21) and form, with c' = sm_scale' * 2^26 (a power-of-two scaling: exact),
 *                   mb = fma(bias, c', m)          rounded ONCE per (row, 64-key tile, k scale)
 *                   P  = exp2(fma(x, c', -mb))     one rounding
 *               Real-number value: s * sm_scale' - m - delta, delta = the rounding of mb (<= half an ulp of |bias * c' + m|): up to
 *               ~0.6 of one INT8 x INT8 score step in the exponent.  The row maximum m itself is formed as in mode 0 (from the exact
 *               integer maximum), in both kernels.  This mode mirrors that rounding for rounding so that the kernel can be held to
 *               2e-3 * max|o| against it; how far it is from mode 0 is a property of the arithmetic, measured on the CPU
 *               (tests/test_oracle_golden.py::test_folded_scores_vs_exact).
 * mask_b / mask_f (at most one non-NULL, [B,Hq,Lq,Lk], non-causal only): attn_mask of the Triton path
 *   (attn_qk_int8_per_block.py:31-51).  bool: a 128x64 tile whose mask block is all False is skipped
 *   (:36-38), otherwise 0 / -1e6 is added to the score (:47-48); float: the value is added (:49-50);
 *   out-of-range keys load as False / -1e6 (`other=`), i.e. their score is exactly -1e6 (K loads as 0).
 */
#define BM 128
#define BN_MAX 128
#define NEG_BIG (-1.0e30f)

/* tile_keys: keys per iteration of the online softmax -- one maximum update, one rescale and one two-level fold per tile.  64 = the sm80 / sm89
 * kernels' CTA_K and the Triton kernels' BLOCK_N (and what the gfx950 kernels run for every entry point); 128 = the sm90 kernel's CTA_K
 * (qk_int_sv_f8_cuda_sm90.cu:127-135,285-356: update_mdo over the 128 keys, RO_temp = P.V over them from zero, RO += RO_temp), with which
 * every P of a tile's FIRST 64 keys is rounded against the maximum over all 128. */
ORC_EXPORT int orc_attn_ex(const int8_t *q, const int8_t *k, const void *v, uint16_t *o, float *lse,
                           const float *q_scale, const int32_t *q_sidx, int nqs,
                           const float *k_scale, const int32_t *k_sidx, int nks,
                           const float *v_scale, const float *v_mean,
                           const uint8_t *mask_b, const float *mask_f,
                           int B, int Hq, int Hkv, int Lq, int Lk, int D,
                           int causal, float c, int pv_mode, int out_dtype, int score_mode, int tile_keys)
{
    if (D > 128 || Hq % Hkv) return -1;
    if (tile_keys != 64 && (tile_keys != 128 || mask_b || mask_f)) return -1;
    const int BN = tile_keys;
    if (score_mode != 0 && (score_mode != 1 || pv_mode < 2 || mask_b || mask_f)) return -1;
    const float bias = bits_f(0x3E22F983u);          /* 1 / (2 pi) as the kernels' MFMA C operand */
    const int g = Hq / Hkv;
    const int nqb = (Lq + BM - 1) / BM;
    const int fp8 = pv_mode >= 2;
    const float off = fp8 ? 8.807f : 0.0f;
    /* V pre-decoded to fp32 once (values are exactly representable) */
    size_t nv = (size_t)B * Hkv * Lk * D;
    float *vf = (float *)malloc(sizeof(float) * nv);
    if (!vf) return -1;
    for (size_t i = 0; i < nv; i++)
        vf[i] = fp8 ? orc_e4m3_2f(((const uint8_t *)v)[i]) : orc_h2f(((const uint16_t *)v)[i]);

#pragma omp parallel for collapse(3) schedule(dynamic)
    for (int b = 0; b < B; b++)
        for (int h = 0; h < Hq; h++)
            for (int qb = 0; qb < nqb; qb++) {
                const int hk = h / g;
                const int8_t *qp = q + ((size_t)(b * Hq + h) * Lq) * D;
                const int8_t *kp = k + ((size_t)(b * Hkv + hk) * Lk) * D;
                const float *vp = vf + ((size_t)(b * Hkv + hk) * Lk) * D;
                const float *qs = q_scale + (size_t)(b * Hq + h) * nqs;
                const float *ks = k_scale + (size_t)(b * Hkv + hk) * nks;
                const int r0 = qb * BM, rows = (Lq - r0 < BM) ? Lq - r0 : BM;
                float m[BM], l[BM];
                float (*acc)[128] = malloc(sizeof(float) * BM * 128);
                float (*p)[BN_MAX] = malloc(sizeof(float) * BM * BN_MAX);
                float tile[128];
                for (int i = 0; i < BM; i++) { m[i] = NEG_BIG; l[i] = 0.0f; }
                memset(acc, 0, sizeof(float) * BM * 128);
                int kend = Lk;
                if (causal && (r0 + BM) < kend) kend = r0 + BM;
                const int masked = (mask_b != NULL) || (mask_f != NULL);
                const size_t mo = ((size_t)(b * Hq + h) * Lq) * Lk;
                for (int n0 = 0; n0 < kend; n0 += BN) {
                    const int nk = (kend - n0 < BN) ? kend - n0 : BN;
                    const int nkv = (Lk - n0 < BN) ? Lk - n0 : BN;   /* keys that exist */
                    if (mask_b) {
                        int any = 0;
                        for (int i = 0; i < rows && !any; i++)
                            for (int j = 0; j < nkv; j++)
                                if (mask_b[mo + (size_t)(r0 + i) * Lk + n0 + j]) { any = 1; break; }
                        if (!any) continue;
                    }
                    for (int i = 0; i < rows; i++) {
                        const int8_t *qr = qp + (size_t)(r0 + i) * D;
                        const float qsc = qs[q_sidx[r0 + i]];
                        float mx = NEG_BIG;       /* non-fused: max score; fused: max of (score - offset) */
                        float dotf[BN_MAX], ccj[BN_MAX];
                        int32_t doti[BN_MAX];
                        /* pv_mode 0 restates the Triton kernel literally: qk = dot * (q_scale*k_scale), then
                         * qk - m (attn_qk_int8_per_block.py:41,53-55).  The other modes restate the CUDA kernels:
                         *   dequant_scale = q_scale * k_scale;  sm_scale' = (sm_scale*log2e) * dequant_scale
                         *                                              (qk_int_sv_f8_cuda_sm89.cuh:263-266,334-335)
                         *   m_temp = fma(max raw score, sm_scale', -offset)      (attn_utils.cuh:372-384)
                         *   P = exp2(fma(raw score, sm_scale', -m))              (attn_utils.cuh:445-449)
                         * A reference thread holds scores of ONE key-scale group: it takes the maximum of the raw
                         * scores, applies its group's FMA, and the results are max-reduced across threads.  FMA with a
                         * non-negative scale is monotone, so max_j fma(raw_j, scale_j, -offset) is the same number. */
                        const int fused = (pv_mode != 0) && !masked;
                        for (int j = 0; j < BN; j++) {
                            float s = NEG_BIG;
                            dotf[j] = 0.0f; ccj[j] = 0.0f;
                            if (j < nkv && j < nk && !(causal && (n0 + j) > (r0 + i))) {
                                const int8_t *kr = kp + (size_t)(n0 + j) * D;
                                int32_t dot = 0;
                                for (int d = 0; d < D; d++) dot += (int32_t)qr[d] * (int32_t)kr[d];
                                doti[j] = dot;
                                if (fused) {
                                    dotf[j] = (float)dot;
                                    ccj[j] = c * (qsc * ks[k_sidx[n0 + j]]);
                                    s = fmaf(dotf[j], ccj[j], -off);
                                } else {
                                    s = (float)dot * (qsc * ks[k_sidx[n0 + j]]) * c;
                                }
                                if (mask_b) s += mask_b[mo + (size_t)(r0 + i) * Lk + n0 + j] ? 0.0f : -1.0e6f;
                                if (mask_f) s += mask_f[mo + (size_t)(r0 + i) * Lk + n0 + j];
                            } else if (masked) s = -1.0e6f;
                            p[i][j] = s;
                            mx = fmaxf(mx, s);
                        }
                        float m_new = fmaxf(m[i], fused ? mx : mx - off);
                        float alpha = exp2f(m[i] - m_new);
                        float rs = 0.0f;
                        for (int j = 0; j < BN; j++) {
                            float e;
                            if (p[i][j] <= NEG_BIG) e = 0.0f;
                            else if (fused && score_mode == 1) {
                                const float c26 = ccj[j] * 67108864.0f;                       /* c' = sm_scale' * 2^26 */
                                const float x = bits_f(0x3E22F983u + (uint32_t)doti[j]);       /* bias + s * 2^-26, exact */
                                const float mb = fmaf(bias, c26, m_new);
                                e = exp2f(fmaf(x, c26, -mb));
                            }
                            else if (fused) e = exp2f(fmaf(dotf[j], ccj[j], -m_new));
                            else e = exp2f(p[i][j] - m_new);
                            p[i][j] = fp8 ? orc_e4m3_2f(orc_f2e4m3(e)) : orc_h2f(orc_f2h(e));
                            rs += (pv_mode == 1) ? p[i][j] : e;   /* sm80 CUDA kernels: tensor-core row sum of the fp16 P */
                        }
                        l[i] = l[i] * alpha + rs;
                        m[i] = m_new;
                        /* P.V for this row */
                        if (pv_mode == 3 || pv_mode == 1) {
                            for (int d = 0; d < D; d++) acc[i][d] *= alpha;
                            for (int j = 0; j < nkv; j++) {
                                const float pj = p[i][j];
                                if (pj == 0.0f) continue;
                                const float *vr = vp + (size_t)(n0 + j) * D;
                                for (int d = 0; d < D; d++) acc[i][d] += pj * vr[d];
                            }
                        } else {
                            for (int d = 0; d < D; d++) tile[d] = 0.0f;
                            for (int j = 0; j < nkv; j++) {
                                const float pj = p[i][j];
                                if (pj == 0.0f) continue;
                                const float *vr = vp + (size_t)(n0 + j) * D;
                                for (int d = 0; d < D; d++) tile[d] += pj * vr[d];
                            }
                            for (int d = 0; d < D; d++) {
                                float t = (pv_mode == 0) ? orc_h2f(orc_f2h(tile[d])) : tile[d];
                                acc[i][d] = acc[i][d] * alpha + t;
                            }
                        }
                    }
                }
                for (int i = 0; i < rows; i++) {
                    uint16_t *orow = o + ((size_t)(b * Hq + h) * Lq + r0 + i) * D;
                    /* a row of a query block whose every tile was skipped (bool mask, all-False blocks): the Triton kernel's l_i starts at 1.0
                     * and nothing multiplies it (attn_qk_int8_per_block.py:111-112,122), so o = 0 / 1 = 0; its lse is log2(1) + (-inf) (:127) */
                    const float li = (l[i] == 0.0f && m[i] == NEG_BIG) ? 1.0f : l[i];
                    for (int d = 0; d < D; d++) {
                        float x = acc[i][d] / li;
                        if (fp8) x *= v_scale[(size_t)(b * Hkv + hk) * D + d];
                        if (v_mean) x += v_mean[(size_t)(b * Hkv + hk) * D + d];   /* sm89.cuh:575-621 */
                        orow[d] = st16(x, out_dtype);
                    }
                    if (lse) lse[(size_t)(b * Hq + h) * Lq + r0 + i] = (l[i] == 0.0f && m[i] == NEG_BIG) ? -INFINITY : log2f(l[i]) + m[i];
                }
                free(acc);
                free(p);
            }
    free(vf);
    return 0;
}

ORC_EXPORT int orc_attn(const int8_t *q, const int8_t *k, const void *v, uint16_t *o, float *lse,
                        const float *q_scale, const int32_t *q_sidx, int nqs,
                        const float *k_scale, const int32_t *k_sidx, int nks,
                        const float *v_scale, const float *v_mean,
                        const uint8_t *mask_b, const float *mask_f,
                        int B, int Hq, int Hkv, int Lq, int Lk, int D,
                        int causal, float c, int pv_mode, int out_dtype, int score_mode)
{
    return orc_attn_ex(q, k, v, o, lse, q_scale, q_sidx, nqs, k_scale, k_sidx, nks, v_scale, v_mean, mask_b, mask_f, B, Hq, Hkv, Lq, Lk, D,
                       causal, c, pv_mode, out_dtype, score_mode, 64);
}

ORC_EXPORT int orc_version(void) { return 3; }
