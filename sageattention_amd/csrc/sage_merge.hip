// sage_merge.hip -- merge of two partial attention states by their log-sum-exp (HBM-bound, elementwise).
//
// The reference exposes `return_lse` (core.py:782-786, 823-826) as its only long-context hook: sequence-parallel
// callers (ring attention, example/parallel_sageattn_cogvideo.py via xfuser) run attention against one K/V shard at
// a time and combine the partial outputs with
//     m = max(lse_a, lse_b);  w_a = e^(lse_a - m);  w_b = e^(lse_b - m)
//     o = (o_a w_a + o_b w_b) / (w_a + w_b);        lse = m + log(w_a + w_b)
// This kernel keeps the running state (o_acc, lse_acc) in FP32 across the steps of the ring so the output is
// rounded to fp16/bf16 exactly once, at the last step.  One thread owns 8 channels of one (batch, head, row):
// 32 B of accumulator + 16 B of new output in, 32 B (+16 B on the last step) out.
#include "sage_common.h"
#include "sage_kernels.h"

namespace sage {

template <int DT>
__global__ void __launch_bounds__(256)
merge_states_kernel(const MergeParams p)
{
    // A row is owned by `cpr_pad` consecutive lanes (cpr rounded up to a power of two <= 64), so all threads of a row
    // sit in ONE wave: every lane's read of lse_acc[row] is issued before the c8 == 0 lane's write of the merged value
    // (wave-lockstep program order).  Spreading a row over waves / workgroups (the round-1 mapping for D = 96, 80, ...)
    // let a later wave read the already-merged lse.
    const int cpr = p.D / 8;                        // 8-channel chunks per row
    const int cpr_pad = p.cpr_pad;
    const long row = (long)blockIdx.x * (256 / cpr_pad) + threadIdx.x / cpr_pad;
    const int c = threadIdx.x % cpr_pad;
    if (row >= (long)p.B * p.H * p.L || c >= cpr) return;
    const int c8 = c * 8;
    const int l = (int)(row % p.L);
    const long bh = row / p.L;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);

    const uint16_t *on = reinterpret_cast<const uint16_t *>(p.o_new) + (long)b * p.n_sb + (long)h * p.n_sh + (long)l * p.n_sl + c8;
    float *acc = p.o_acc + row * p.D + c8;
    const v4u raw = *reinterpret_cast<const v4u *>(on);
    float xn[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned w = raw[j >> 1];
        xn[j] = ld16<DT>((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
    }
    const float ln = p.lse_new[row];
    float r[8], lse;
    if (p.first) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = xn[j];
        lse = ln;
    } else {
        const float la = p.lse_acc[row];
        const float m = fmaxf(la, ln);
        if (m == -INFINITY) {                       // neither side attended to anything: keep the (zero) state
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = 0.0f;
            lse = -INFINITY;
        } else {
            const float wa = __expf(la - m), wb = __expf(ln - m);
            const float s = wa + wb, inv = 1.0f / s;
            const v4f a0 = *reinterpret_cast<const v4f *>(acc), a1 = *reinterpret_cast<const v4f *>(acc + 4);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = ((j < 4 ? a0[j] : a1[j - 4]) * wa + xn[j] * wb) * inv;
            lse = m + __logf(s);
        }
    }
    v4f o0 = {r[0], r[1], r[2], r[3]}, o1 = {r[4], r[5], r[6], r[7]};
    *reinterpret_cast<v4f *>(acc) = o0;
    *reinterpret_cast<v4f *>(acc + 4) = o1;
    if (c8 == 0) p.lse_acc[row] = lse;
    if (p.o_out != nullptr) {
        uint16_t *oo = reinterpret_cast<uint16_t *>(p.o_out) + (long)b * p.o_sb + (long)h * p.o_sh + (long)l * p.o_sl + c8;
        v4u pk;
#pragma unroll
        for (int w = 0; w < 4; w++) pk[w] = (unsigned)st16<DT>(r[2 * w]) | ((unsigned)st16<DT>(r[2 * w + 1]) << 16);
        *reinterpret_cast<v4u *>(oo) = pk;
    }
}

// ---- split-KV merge: S partial states of one attention call -> final output (one pass) ---------------------------------
// The partial outputs are fp16 (11-bit significand: the merge adds no visible rounding for fp16 / bf16 results) in the
// order the attention kernel wrote them with the KV chunks folded into the kv-head dimension: [B, Hkv, S, group, L, D]
// (query head h = hk * group + g), and their log-sum-exps (log2 domain, as the kernel leaves them) [B, Hkv, S, group, L];
// an optional (S+1)-th chunk (a ragged tail of the key range computed by a second launch) comes as [B, H, L, D] / [B, H, L].
// One thread owns 8 channels of one row.
template <int DT>
__global__ void __launch_bounds__(256)
merge_split_kernel(const SplitMergeParams p)
{
    const int cpr = p.D / 8;
    const int cpr_pad = p.cpr_pad;
    const long row = (long)blockIdx.x * (256 / cpr_pad) + threadIdx.x / cpr_pad;
    const int c = threadIdx.x % cpr_pad;
    if (row >= (long)p.B * p.H * p.L || c >= cpr) return;
    const int c8 = c * 8;
    const int l = (int)(row % p.L);
    const long bh = row / p.L;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const int hk = h / p.group, gq = h - hk * p.group;
    const long HL = (long)p.group * p.L;                                              // chunk stride in rows
    const long r0 = (((long)b * (p.H / p.group) + hk) * p.S) * HL + (long)gq * p.L + l;   // row of chunk 0
    const float *la = p.lse_part + r0;                                                // + s * HL
    const uint16_t *oa = reinterpret_cast<const uint16_t *>(p.o_part) + r0 * p.D + c8;
    const bool tail = p.o_tail != nullptr;
    const float lt = tail ? p.lse_tail[row] : -INFINITY;
    float m = lt;
    for (int s = 0; s < p.S; s++) m = fmaxf(m, la[(long)s * HL]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wsum = 0.0f;
    auto add = [&](const uint16_t *src, float lse) {
        if (lse == -INFINITY) return;                       // chunk with no visible key for this row
        const float w = __builtin_amdgcn_exp2f(lse - m);
        const v4u raw = *reinterpret_cast<const v4u *>(src);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned wd = raw[j >> 1];
            acc[j] += w * f16_to_f32((uint16_t)((j & 1) ? (wd >> 16) : (wd & 0xffffu)));
        }
        wsum += w;
    };
    if (m != -INFINITY) {
        for (int s = 0; s < p.S; s++) add(oa + (long)s * HL * p.D, la[(long)s * HL]);
        if (tail) add(reinterpret_cast<const uint16_t *>(p.o_tail) + row * p.D + c8, lt);
    }
    const float inv = wsum > 0.0f ? 1.0f / wsum : 0.0f;
    uint16_t *oo = reinterpret_cast<uint16_t *>(p.o_out) + (long)b * p.o_sb + (long)h * p.o_sh + (long)l * p.o_sl + c8;
    v4u pk;
#pragma unroll
    for (int w = 0; w < 4; w++) pk[w] = (unsigned)st16<DT>(acc[2 * w] * inv) | ((unsigned)st16<DT>(acc[2 * w + 1] * inv) << 16);
    *reinterpret_cast<v4u *>(oo) = pk;
    if (p.lse_out != nullptr && c8 == 0) p.lse_out[row] = wsum > 0.0f ? m + __builtin_amdgcn_logf(wsum) : -INFINITY;   // v_log_f32 = log2
}

hipError_t launch_merge_split(const SplitMergeParams &p, hipStream_t stream)
{
    const long rows = (long)p.B * p.H * p.L;
    if (rows <= 0) return hipSuccess;
    SplitMergeParams q = p;
    q.cpr_pad = 1;
    while (q.cpr_pad < p.D / 8) q.cpr_pad *= 2;
    if (q.cpr_pad > 64) return hipErrorInvalidValue;
    const long rpb = 256 / q.cpr_pad;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    if (p.dtype == DT_F16) hipLaunchKernelGGL(merge_split_kernel<DT_F16>, grid, dim3(256), 0, stream, q);
    else hipLaunchKernelGGL(merge_split_kernel<DT_BF16>, grid, dim3(256), 0, stream, q);
    return hipGetLastError();
}

hipError_t launch_merge_states(const MergeParams &p, hipStream_t stream)
{
    const long rows = (long)p.B * p.H * p.L;
    if (rows <= 0) return hipSuccess;
    MergeParams q = p;
    q.cpr_pad = 1;
    while (q.cpr_pad < p.D / 8) q.cpr_pad *= 2;
    if (q.cpr_pad > 64) return hipErrorInvalidValue;          // head_dim <= 512 (checked by the C ABI)
    const long rpb = 256 / q.cpr_pad;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    if (p.dtype == DT_F16) hipLaunchKernelGGL(merge_states_kernel<DT_F16>, grid, dim3(256), 0, stream, q);
    else hipLaunchKernelGGL(merge_states_kernel<DT_BF16>, grid, dim3(256), 0, stream, q);
    return hipGetLastError();
}

}  // namespace sage
