// sage_prep_v.hip -- V pre-pass: per-channel FP8 quantisation (or fp16 pass-through) fused with
// the transpose into the gfx950 PV-operand tile image.
//
// Replaces the reference's three-kernel V path
//   csrc/fused/fused.cu:262-313  TransposePadPermuteKernel ([L,D] -> [D, ceil64(L)], NVIDIA order)
//   csrc/fused/fused.cu:316-427  MeanScaleKernel           (per-(b,h,d) amax, scale, e4m3 cast)
//   sageattention/quant.py:224-293 per_channel_fp8          (host wrapper)
// and, for the FP16-PV paths, `v.to(torch.float16)` (core.py:297-298,613).
//
// Tile image (one per 64 tokens of one (batch, kv-head)), D rows:
//   fp8 : row d = 64 bytes, byte (16*ch' + i) holds token tau(16*ch + i), ch' = ch ^ ((d>>2)&3)
//   fp16: row d = 128 bytes, elem (8*ch' + i) holds token tau( 8*ch + i), ch' = ch ^ ((d>>1)&7)
// with tau() = sage::pv_token_of_position.  The image is byte-for-byte what the attention
// kernel keeps in LDS, so the tile load is a linear 16-B-per-lane copy and the PV A-operand
// reads are conflict-free ds_read_b128.  Tokens >= L are zero (the reference zero-pads too,
// fused.cu:283-286) so masked probabilities never multiply garbage.
//
// HBM traffic: the statistics pass (sage_stats.hip) reads V once (2 B/elt); the quantise pass reads it
// again and writes 1 B/elt (fp8) -- 5 B/elt against the reference's 9 B/elt (fp16 intermediate written + read twice).
#include "sage_common.h"
#include "sage_kernels.h"

namespace sage {

template <int D, int DT, bool FP8>
__global__ void __launch_bounds__(256)
prep_v_kernel(const PrepVParams p)
{
    constexpr int LDT = D + 8;                                  // padded row (16-B aligned)
    __shared__ __attribute__((aligned(16))) uint16_t tile[BLKK * LDT];
    const int tid = threadIdx.x;
    const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    int L = p.L;
    long voff, tile_idx;
    if (p.cu != nullptr) {
        const int s0 = p.cu[b];
        L = p.cu[b + 1] - s0;
        if (t * BLKK >= L) return;
        voff = (long)s0 * p.v_sl + (long)h * p.v_sh;
        tile_idx = ((long)p.cu_tiles[b] + t) * p.H + h;
    } else {
        voff = (long)b * p.v_sb + (long)h * p.v_sh;
        tile_idx = ((long)b * p.H + h) * ((L + BLKK - 1) / BLKK) + t;
    }
    const uint16_t *v = reinterpret_cast<const uint16_t *>(p.v) + voff;

    constexpr int TPR = D / 8;
#pragma unroll
    for (int i = 0; i < BLKK * TPR / 256; i++) {
        const int piece = tid + 256 * i;
        const int row = piece / TPR, c8 = (piece % TPR) * 8;
        const int tok = t * BLKK + row;
        v4u raw = {0u, 0u, 0u, 0u};
        if (tok < L) raw = *reinterpret_cast<const v4u *>(v + (long)tok * p.v_sl + c8);
        *reinterpret_cast<v4u *>(&tile[row * LDT + c8]) = raw;
    }
    __syncthreads();

    if constexpr (FP8) {
        // per-channel scale (and mean for smooth_v) from the (max, min, sum) statistics:
        //   plain   : amax = max(|max|, |min|)                              (fused.cu:388)
        //   smooth_v: mean = sum / ceil16(L)  (the reference divides by the 16-padded length,
        //             fused.cu:335,381), amax = max(|max - mean|, |min - mean|)  (fused.cu:383-385)
        //   The reference takes max/min over the ceil16(L) tokens of its zero-padded transpose (fused.cu:335-357), so
        //   for L % 16 != 0 the padding zeros take part: irrelevant for the plain amax, part of the smooth_v amax.
        const float *st = p.stats + ((long)b * p.H + h) * 3 * D;
        const bool smooth = p.v_mean != nullptr;
        const float lpad = (float)((L + 15) / 16 * 16);
        const bool padded = (L & 15) != 0;
        auto chan = [&](int d, float &mean, float &am) {
            mean = smooth ? st[2 * D + d] / lpad : 0.0f;
            float mx = st[d], mn = st[D + d];
            if (padded) { mx = fmaxf(mx, 0.0f); mn = fminf(mn, 0.0f); }
            am = fmaxf(fabsf(mx - mean), fabsf(mn - mean));
        };
        if (t == 0 && tid < D) {
            float mean, am;
            chan(tid, mean, am);
            p.v_scale[((long)b * p.H + h) * D + tid] = am / p.scale_max;
            if (smooth) p.v_mean[((long)b * p.H + h) * D + tid] = mean;
        }
        unsigned char *out = reinterpret_cast<unsigned char *>(p.out) + tile_idx * (long)(D * 64);
#pragma unroll
        for (int i = 0; i < D * 4 / 256; i++) {
            const int piece = tid + 256 * i;
            const int d = piece >> 2, pc = piece & 3;
            const int ch = swz_chunk<64>(d, pc);                   // involution: physical <-> logical
            float mean, am;
            chan(d, mean, am);
            const float recp = am > 0.0f ? p.scale_max / am : 0.0f;   // fused.cu:395
            float f[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int tok = pv_token_of_position(16 * ch + j);
                float x = ld16<DT>(tile[tok * LDT + d]);
                if (smooth) x = (t * BLKK + tok < L) ? x - mean : 0.0f;   // padding stays zero
                x *= recp;
                f[j] = fminf(fmaxf(x, -448.0f), 448.0f);             // satfinite
            }
            v4u pk;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                int word = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w], f[4 * w + 1], 0, false);
                word = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w + 2], f[4 * w + 3], word, true);
                pk[w] = (unsigned)word;
            }
            __builtin_nontemporal_store(pk, reinterpret_cast<v4u *>(out + d * 64 + pc * 16));
        }
    } else {
        unsigned char *out = reinterpret_cast<unsigned char *>(p.out) + tile_idx * (long)(D * 128);
#pragma unroll
        for (int i = 0; i < D * 8 / 256; i++) {
            const int piece = tid + 256 * i;
            const int d = piece >> 3, pc = piece & 7;
            const int ch = swz_chunk<128>(d, pc);
            v4u pk;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                unsigned word = 0;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int tok = pv_token_of_position(8 * ch + 2 * w + e);
                    uint16_t raw = tile[tok * LDT + d];
                    if (p.mean_in != nullptr) {   // sub_mean (quant.py:182-222, fused.cu:200-260): (v - vm) -> fp16
                        const float m = p.mean_in[((long)b * p.H + h) * D + d];
                        raw = (t * BLKK + tok < L) ? f32_to_f16_rne(ld16<DT>(raw) - m) : (uint16_t)0;
                    } else if (DT != DT_F16) raw = f32_to_f16_rne(bf16_to_f32(raw));   // v.to(float16)
                    word |= (unsigned)raw << (16 * e);
                }
                pk[w] = word;
            }
            __builtin_nontemporal_store(pk, reinterpret_cast<v4u *>(out + d * 128 + pc * 16));
        }
    }
}

hipError_t launch_prep_v(const PrepVParams &p, hipStream_t s)
{
    const int nt = (p.L + BLKK - 1) / BLKK;               // varlen: p.L = max_seqlen
    if (nt <= 0 || p.B <= 0) return hipSuccess;
    dim3 grid(nt, p.H, p.B);
#define SAGE_PV(D_, T_, F_) hipLaunchKernelGGL((prep_v_kernel<D_, T_, F_>), grid, dim3(256), 0, s, p)
    if (p.D == 128) {
        if (p.fp8) { if (p.dtype == DT_F16) SAGE_PV(128, DT_F16, true); else SAGE_PV(128, DT_BF16, true); }
        else       { if (p.dtype == DT_F16) SAGE_PV(128, DT_F16, false); else SAGE_PV(128, DT_BF16, false); }
    } else if (p.D == 64) {
        if (p.fp8) { if (p.dtype == DT_F16) SAGE_PV(64, DT_F16, true); else SAGE_PV(64, DT_BF16, true); }
        else       { if (p.dtype == DT_F16) SAGE_PV(64, DT_F16, false); else SAGE_PV(64, DT_BF16, false); }
    } else return hipErrorInvalidValue;
#undef SAGE_PV
    return hipGetLastError();
}

}  // namespace sage
