// sage_stats.hip -- deterministic per-channel statistics over the sequence (HBM-bound, one read).
//
// One kernel pair serves every reduction of the pre-pass:
//   * K smoothing mean   km = k.mean(dim=seq)                      (core.py:280; a torch reduction there)
//   * V per-channel amax / min / max (/ sum for smooth_v)           (MeanScaleKernel pass 1, fused.cu:345-381)
// Stage 1: grid (slab, head, batch); a workgroup reduces SLAB tokens x D channels to per-channel
//          (max, min, sum) partials, written to a workspace [B,H,nslab,3,D] -- no atomics, so the
//          result is bit-reproducible run to run (the parity tests rely on that).
// Stage 2: grid (head, batch); reduces the slabs in index order into stats[B,H,3,D] and optionally
//          emits the mean in the input dtype (the K-smoothing mean).
#include "sage_common.h"
#include "sage_kernels.h"

namespace sage {

template <int D, int DT>
__global__ void __launch_bounds__(256)
stats_partial_kernel(const StatsParams p)
{
    constexpr int TPR = D / 8;           // threads per row (16 B each)
    constexpr int RPI = 256 / TPR;       // rows per iteration
    __shared__ float red[3][RPI][D];
    const int tid = threadIdx.x;
    const int slab = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const uint16_t *x = reinterpret_cast<const uint16_t *>(p.x) + (long)b * p.x_sb + (long)h * p.x_sh;
    const int c8 = (tid % TPR) * 8, r0 = tid / TPR;
    float mx[8], mn[8], sm[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { mx[j] = -INFINITY; mn[j] = INFINITY; sm[j] = 0.0f; }
    // rows [start, end) of the slab.  Packed batches (p.slab_seq): a slab is 512 tokens of ONE sequence -- the partition of the one-launch
    // varlen pre-pass (sage_prepass.hip), so that both routes sum in the same order and give the same mean bit for bit
    int start = slab * kStatsSlab, end = min(p.L, slab * kStatsSlab + kStatsSlab);
    if (p.slab_seq != nullptr) {
        if (slab >= p.hdr[4]) return;
        const int seg = p.slab_seq[slab];
        int t0, len;
        varlen_segment(p.cu, p.nseq, p.L, seg, t0, len);
        start = t0 + (slab - p.slab_first[seg]) * kStatsSlab;
        end = min(t0 + len, start + kStatsSlab);
    }
    // eight independent 16-byte loads in flight per thread: with one load per thread the 1024-workgroup grid keeps
    // only ~4 MB in flight chip-wide, half of what a cold HBM read stream needs; rows are still accumulated in order
    constexpr int UNR = 8;
    for (int r = start + r0; r < end; r += UNR * RPI) {
        v4u raw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int rr = r + u * RPI;
            raw[u] = *reinterpret_cast<const v4u *>(x + (long)(rr < end ? rr : r) * p.x_sl + c8);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (r + u * RPI < end) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const unsigned w = raw[u][j >> 1];
                    const float f = ld16<DT>((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
                    mx[j] = fmaxf(mx[j], f);
                    mn[j] = fminf(mn[j], f);
                    sm[j] += f;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { red[0][r0][c8 + j] = mx[j]; red[1][r0][c8 + j] = mn[j]; red[2][r0][c8 + j] = sm[j]; }
    __syncthreads();
    if (tid < D) {
        float a = -INFINITY, c = INFINITY, s = 0.0f;
#pragma unroll
        for (int r = 0; r < RPI; r++) { a = fmaxf(a, red[0][r][tid]); c = fminf(c, red[1][r][tid]); s += red[2][r][tid]; }
        float *ws = p.ws + (((long)b * p.H + h) * p.nslab + slab) * 3 * D;
        ws[tid] = a; ws[D + tid] = c; ws[2 * D + tid] = s;
    }
}

template <int D, int DT>
__global__ void stats_final_kernel(const StatsParams p)
{
    const int d = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const float *ws = p.ws + ((long)b * p.H + h) * p.nslab * 3 * D;
    const int nslab = p.slab_seq != nullptr ? p.hdr[4] : p.nslab;      // packed batches: p.nslab is the host-known bound (workspace stride)
    float a = -INFINITY, c = INFINITY, s = 0.0f;
    // slabs in index order (the summation order every route shares); sixteen slabs' loads are in flight together -- issued one by one
    // this loop is a chain of L2 round trips: 19 us for the 66 slabs of a C4 call
    constexpr int NB = 16;
    for (int i0 = 0; i0 < nslab; i0 += NB) {
        float va[NB], vc[NB], vs[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const float *wi = ws + (long)(i0 + u < nslab ? i0 + u : nslab - 1) * 3 * D;
            va[u] = wi[d]; vc[u] = wi[D + d]; vs[u] = wi[2 * D + d];
        }
#pragma unroll
        for (int u = 0; u < NB; u++)
            if (i0 + u < nslab) { a = fmaxf(a, va[u]); c = fminf(c, vc[u]); s += vs[u]; }
    }
    if (p.stats != nullptr) {
        float *st = p.stats + ((long)b * p.H + h) * 3 * D;
        st[d] = a; st[D + d] = c; st[2 * D + d] = s;
    }
    if (p.mean_out != nullptr)      // k.mean(dim=seq) in the input dtype: fp32 sum / L, one rounding
        reinterpret_cast<uint16_t *>(p.mean_out)[((long)b * p.H + h) * D + d] = st16<DT>(s / (float)p.L);
}

hipError_t launch_stats(const StatsParams &p, hipStream_t s)
{
    if (p.B <= 0 || p.H <= 0 || p.nslab <= 0) return hipSuccess;
    dim3 g1(p.nslab, p.H, p.B), g2(p.H, p.B);
#define SAGE_ST(D_, T_) do { hipLaunchKernelGGL((stats_partial_kernel<D_, T_>), g1, dim3(256), 0, s, p); \
                             hipLaunchKernelGGL((stats_final_kernel<D_, T_>), g2, dim3(D_), 0, s, p); } while (0)
    if (p.D == 128) { if (p.dtype == DT_F16) SAGE_ST(128, DT_F16); else SAGE_ST(128, DT_BF16); }
    else if (p.D == 64) { if (p.dtype == DT_F16) SAGE_ST(64, DT_F16); else SAGE_ST(64, DT_BF16); }
    else return hipErrorInvalidValue;
#undef SAGE_ST
    return hipGetLastError();
}

}  // namespace sage
