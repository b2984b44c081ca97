// sage_varlen_plan.hip -- every index array a packed-batch (varlen) call needs, built on the device in ONE small launch, so that
// sageattn_varlen never synchronises with the host (the reference sizes its scale tensors with .item(), quant_per_block_varlen.py:75-76,
// and builds the prefix arrays with torch ops, :68-73):
//   cu_qs / cu_ks   exclusive prefix sums of ceil(Lq_i / blkq), ceil(Lk_i / blkk)   (scale-block and V-tile prefixes)
//   order           the sequences by descending query length                       (legacy unit order of the attention launcher)
//   items / hdr     the work list of the attention launch: every existing (sequence, 128-row query block) pair, sorted by descending
//                   weight (csrc/sage_work_order.h), and the launch plan over it {nitems, group, fold, left, nslab, max Lk, sum Lk}
//   slab_first / slab_seq   the 512-token slabs of the K / V pre-pass, per sequence (prefix sums and slab -> sequence map), plus the slabs of
//                   the rows outside every sequence (two gap segments), which only the K mean reads
// One workgroup: nseq <= kVarlenPlanMaxSeq; Hillis-Steele scans in LDS, ranks by the closed-form counts of sage_work_order.h.
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_work_order.h"

namespace sage {

__global__ void __launch_bounds__(1024) varlen_plan_kernel(const VarlenPlanParams p)
{
    __shared__ int sq[2][kVarlenPlanMaxSeq], sk[2][kVarlenPlanMaxSeq], ss[2][kVarlenPlanMaxSeq];
    __shared__ int lq_s[kVarlenPlanMaxSeq], lk_s[kVarlenPlanMaxSeq];
    __shared__ int max_lk;
    const int i = threadIdx.x, nseq = p.nseq;
    int lq = 0, lk = 0;
    if (i == 0) max_lk = 0;
    __syncthreads();
    if (i < nseq) {
        lq = p.cu_q[i + 1] - p.cu_q[i];
        lk = p.cu_k[i + 1] - p.cu_k[i];
        lq = lq > 0 ? lq : 0;
        lk = lk > 0 ? lk : 0;
        sq[0][i] = (lq + p.blkq - 1) / p.blkq;
        sk[0][i] = (lk + p.blkk - 1) / p.blkk;
        ss[0][i] = (lk + kStatsSlab - 1) / kStatsSlab;
        lq_s[i] = lq;
        lk_s[i] = lk;
        atomicMax(&max_lk, lk);
    }
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < nseq; d <<= 1) {                 // inclusive scans
        if (i < nseq) {
            sq[cur ^ 1][i] = sq[cur][i] + (i >= d ? sq[cur][i - d] : 0);
            sk[cur ^ 1][i] = sk[cur][i] + (i >= d ? sk[cur][i - d] : 0);
            ss[cur ^ 1][i] = ss[cur][i] + (i >= d ? ss[cur][i - d] : 0);
        }
        cur ^= 1;
        __syncthreads();
    }
    if (i < nseq) {
        if (p.cu_qs != nullptr) p.cu_qs[i + 1] = sq[cur][i];
        p.cu_ks[i + 1] = sk[cur][i];
        if (p.slab_first != nullptr) p.slab_first[i + 1] = ss[cur][i];
        if (p.order != nullptr) {
            int rank = 0;                                // longer sequences first; ties by index (a total order: every slot written once)
            for (int j = 0; j < nseq; j++) rank += (lq_s[j] > lq || (lq_s[j] == lq && j < i)) ? 1 : 0;
            p.order[rank] = i;
        }
    }
    // two more slab-owning segments, read by the K mean only: the rows behind the last sequence and in front of the first one
    const int nslab_seq = ss[cur][nseq - 1];
    const int tail = max(p.total_k - p.cu_k[nseq], 0), head = max(p.cu_k[0], 0);
    const int nslab_tail = (tail + kStatsSlab - 1) / kStatsSlab, nslab_head = (head + kStatsSlab - 1) / kStatsSlab;
    // the counts come from cu_seqlens on the device, the buffers were sized on the host from q.shape[0] / k.shape[0]: a cu_seqlens that is
    // inconsistent with them (last prefix beyond the rows, not monotonic) must not make this kernel -- or the launches that read hdr -- run
    // past the allocations.  Clamped counts drop work (the call's result is then as undefined as its input), they never write out of bounds.
    const int nitems_all = sq[cur][nseq - 1], nslab_all = nslab_seq + nslab_tail + nslab_head;
    const int nitems = p.items != nullptr ? min(nitems_all, p.items_cap) : nitems_all;
    const int nslab = p.slab_seq != nullptr ? min(nslab_all, p.slab_cap) : nslab_all;
    if (i == 0) {
        if (p.cu_qs != nullptr) p.cu_qs[0] = 0;
        p.cu_ks[0] = 0;
        if (p.slab_first != nullptr) { p.slab_first[0] = 0; p.slab_first[nseq + 1] = nslab_seq + nslab_tail; p.slab_first[nseq + 2] = nslab; }
        if (p.hdr != nullptr) {
            WorkOrder w;
            plan_varlen_order(w, p.Hq, p.Hq / p.Hkv, nitems, (long)max_lk, p.head_dim, p.pv_fp8 != 0, p.forced_group);
            p.hdr[0] = nitems; p.hdr[1] = w.group; p.hdr[2] = w.fold; p.hdr[3] = w.left;
            p.hdr[4] = nslab; p.hdr[5] = max_lk; p.hdr[6] = p.cu_k[nseq] - p.cu_k[0]; p.hdr[7] = max(nslab - nslab_seq, 0);
        }
    }
    // inclusive scan value of sequence t = number of blocks / slabs in sequences 0 .. t: the first t with scan[t] > idx owns index idx
    auto owner = [&](const int *scan, int idx) {
        int lo = 0, hi = nseq - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (scan[mid] > idx) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    if (p.items != nullptr && p.blkq == BLKQ && p.blkk == BLKK) {
        for (int idx = i; idx < nitems_all; idx += 1024) {
            const int s = owner(sq[cur], idx);
            const int j = idx - (s > 0 ? sq[cur][s - 1] : 0);
            const int rank = varlen_item_rank(lq_s, lk_s, nseq, s, j, p.causal != 0);
            if (rank >= 0 && rank < nitems) {
                p.items[2 * rank] = s;
                p.items[2 * rank + 1] = j;
            }
        }
    }
    if (p.slab_seq != nullptr) {
        for (int idx = i; idx < nslab; idx += 1024)
            p.slab_seq[idx] = idx < nslab_seq ? owner(ss[cur], idx) : (idx < nslab_seq + nslab_tail ? nseq : nseq + 1);
    }
}

hipError_t launch_varlen_plan(const VarlenPlanParams &p, hipStream_t stream)
{
    if (p.nseq <= 0) return hipSuccess;
    if (p.nseq > kVarlenPlanMaxSeq) return hipErrorInvalidValue;
    hipLaunchKernelGGL(varlen_plan_kernel, dim3(1), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

}  // namespace sage
