// sage_work_order.h -- which (head, query block) item a workgroup of a causal dense attention launch takes.
// One definition for the kernel (device) and for the launcher and the CPU tests (host, through the C ABI's sage_debug_work_* entry
// points: tests/test_work_order.py checks THIS code, not a restatement).  The reference leaves the order to the hardware: blockIdx.x =
// query block, ascending (csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:720-738).
//
// What the order is built on (tools/microbench/ubench7_dispatch.hip, profiles/r3_run_j_ubench7_dispatch.txt): gfx950 deals blockIdx
// round-robin to its 8 XCDs; inside an XCD the first 32 indices land one per CU and the next 32 on the same CUs again (in-XCD indices i
// and i + 32 share a CU); later workgroups go to whichever slot frees first, in index order.
//
// B * Hq = 8 * hpx + left heads.  Every XCD owns hpx whole heads (their K/V stream through one L2); the `left` heads are dealt to all
// eight XCDs by query block, in octets of blocks with alternating direction, so every XCD gets the same mix of long and short blocks.
// An XCD's list: the left-over heads' blocks first, then its own heads in groups of `group`; inside a group the longest query block
// comes first ACROSS the heads, so the list ends on the shortest blocks of several heads instead of on one head's longest.  `fold`: the
// grid fits the XCD's resident slots in one round; index i >= 32 takes the (i - 32)-th item from the END, so a CU's two workgroups are
// the i-th longest and the i-th shortest block.
#pragma once

#if defined(__HIPCC__)
#define SAGE_HD __host__ __device__ __forceinline__
#else
#define SAGE_HD inline
#endif

namespace sage {

struct WorkOrder {
    int group;      // heads per group; 0 = head-major contiguous runs (rounds 1-2; non-causal, masked, split-KV launches)
    int fold;       // 1: single-round grid, long + short block per CU
    int left;       // (B * Hq) % 8 heads dealt to all XCDs by query block
};

// blockIdx `bid` of a grid of `nwg` workgroups -> (head = b * Hq + h, rank of the query block: 0 = the last = longest one).
// false: the workgroup has no item (ragged octets of the left-over heads).
SAGE_HD bool work_item(const WorkOrder &w, int bid, int nwg, int nheads, int nqblk, int &head, int &qrank)
{
    const int xcd = bid & 7, idx = bid >> 3;
    const int qq = nwg >> 3, rr = nwg & 7;
    if (w.group <= 0) {     // one contiguous run of the head-major list per XCD, longest block of each head first
        const int wid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        head = wid / nqblk;
        qrank = wid - head * nqblk;
        return true;
    }
    int r = idx;
    if (w.fold != 0 && idx >= 32) r = qq - 1 - (idx - 32);
    const int hpx = nheads >> 3;
    const int left_cnt = w.left * ((nqblk + 7) >> 3);
    if (r < left_cnt) {
        const int oct = r / w.left;
        head = r - oct * w.left;
        qrank = 8 * oct + ((oct & 1) ? 7 - xcd : xcd);
        return qrank < nqblk;
    }
    r -= left_cnt;
    const int gsz = w.group * nqblk;
    const int gi = r / gsz, within = r - gi * gsz;
    const int rest = hpx - gi * w.group;
    const int gc = rest < w.group ? rest : w.group;
    qrank = within / gc;
    head = w.left + xcd * hpx + gi * w.group + (within - qrank * gc);
    return true;
}

// Launcher side: the order of a causal dense launch and its grid size.  `forced`: -1 automatic, n > 0 groups of n heads.
//   balance: the last group's work has to cover its own longest block on all resident slots, G * nqblk * (nqblk + 1) / slots >=
//            2 * nqblk, or the launch ends on a tail of one head's long blocks (C2, N=4k: G=2 988, G=4 1085 TFLOP/s);
//   L2:      the G heads of a group stream their K/V at the same time; past the XCD's 4 MB L2 every further head is re-fetched
//            (C3, 2 MB per head: FETCH_SIZE 105 k KiB at G=1, 148 k at G=2, 284 k at G=4, 587 k at G=8 for 7.4 / 8.9 / 9.3 % less
//            time than head-major), so up to twice the balance size is taken only while the group fits the L2
// (profiles/r3_run_j_work_order_ab.txt, r3_run_k_order_traffic.txt).
SAGE_HD int plan_work_order(WorkOrder &w, int nheads, int nqblk, long kv_len, int head_dim, bool pv_fp8, int forced)
{
    const int hpx = nheads / 8, left = nheads % 8;
    const int wg_per_cu = head_dim == 64 ? 3 : 2;                 // SAGE_MIN_WAVES of the 128-row kernels
    const int slots = 32 * wg_per_cu;
    const int g_bal = (2 * slots + nqblk) / (nqblk + 1);
    const long head_bytes = kv_len * head_dim * (pv_fp8 ? 2 : 3);              // INT8 K + FP8 / FP16 V image
    const int g_l2 = (int)((4L << 20) / (head_bytes > 0 ? head_bytes : 1));
    const int g_more = 2 * g_bal < g_l2 ? 2 * g_bal : g_l2;
    int grp = forced > 0 ? forced : (g_bal > g_more ? g_bal : g_more);
    grp = grp > hpx ? hpx : grp;
    grp = grp < 1 ? 1 : grp;
    w.group = grp;
    w.left = left;
    const int cnt = left * ((nqblk + 7) / 8) + hpx * nqblk;
    const bool one_sorted_list = (left == 0 && grp >= hpx) || hpx == 0;
    w.fold = (wg_per_cu == 2 && cnt > 32 && cnt <= 64 && one_sorted_list) ? 1 : 0;
    return 8 * cnt;
}

// ---- packed (varlen) batches ---------------------------------------------------------------------------------------------------
// The query blocks of ALL sequences form one item list per query head, sorted by descending weight (64-key tiles the block visits);
// the launch then IS a dense launch over nheads = Hq heads of `nitems` "query blocks" each: the same work_item() deals the list to the
// XCDs, so a kv-head's K/V of every sequence streams through one L2 and the heaviest blocks are dispatched first.  The list is built on
// the device (varlen_plan_kernel, sage_quant.hip) from the functions below; sage_debug_varlen_items runs the same functions on the host.
// The reference sizes its grid by max_seqlen_q for every sequence and lets the blocks past a sequence's end exit
// (triton/attn_qk_int8_block_varlen.py:22-30,98-121).

// 64-key tiles query block j (128 rows) of a sequence with lq queries and lk keys visits (top-left causal mask: key <= query)
SAGE_HD int varlen_item_weight(int lk, int j, bool causal)
{
    const int ntk = (lk + 63) >> 6;
    if (!causal) return ntk;
    const int lim = 2 * j + 2;
    return lim < ntk ? lim : ntk;
}

// number of query blocks j in [0, nq) of that sequence whose weight is > w
SAGE_HD int varlen_count_heavier(int nq, int lk, int w, bool causal)
{
    const int ntk = (lk + 63) >> 6;
    if (ntk <= w || nq <= 0) return 0;
    if (!causal) return nq;
    const int first = w < 0 ? 0 : (w >> 1);            // 2 j + 2 > w  <=>  j >= floor(w / 2)
    return first < nq ? nq - first : 0;
}

// rank of item (sequence s, query block j) in the list sorted by (weight descending, sequence index ascending, query block descending):
// a total order, so every rank in [0, nitems) is taken exactly once.  lq / lk: the nseq sequence lengths.
SAGE_HD int varlen_item_rank(const int *lq, const int *lk, int nseq, int s, int j, bool causal)
{
    const int w = varlen_item_weight(lk[s], j, causal);
    int rank = 0;
    for (int t = 0; t < nseq; t++) {
        const int nq = (lq[t] + 127) >> 7;
        const int gt = varlen_count_heavier(nq, lk[t], w, causal);
        rank += gt;
        if (t < s) rank += varlen_count_heavier(nq, lk[t], w - 1, causal) - gt;           // equal weight, earlier sequence
    }
    // equal weight, same sequence, later query block (weights do not decrease with j, so these are j + 1 .. last block of weight w)
    const int nq_s = (lq[s] + 127) >> 7;
    const int ge_w = varlen_count_heavier(nq_s, lk[s], w - 1, causal);                      // blocks of weight >= w: the last ge_w blocks
    const int gt_w = varlen_count_heavier(nq_s, lk[s], w, causal);
    rank += (nq_s - gt_w) - 1 - j;                                                         // blocks j' with j < j' < nq_s - gt_w
    (void)ge_w;
    return rank;
}

// the launch plan over that list: group / fold / left as for a dense causal launch of `nheads` = Hq heads with `nitems` blocks each,
// the group rounded up to whole GQA groups (their query heads share one K/V stream at no L2 cost); returns the grid size
// (`forced`: -1 automatic, n > 0 groups of n heads -- sage_set_work_order / SAGE_ORDER_GROUP, experiments)
SAGE_HD int plan_varlen_order(WorkOrder &w, int nheads, int gqa_group, int nitems, long max_kv_len, int head_dim, bool pv_fp8, int forced = -1)
{
    if (nitems <= 0) { w.group = 1; w.fold = 0; w.left = nheads & 7; return 0; }
    const int grid = plan_work_order(w, nheads, nitems, max_kv_len, head_dim, pv_fp8, forced > 0 ? forced : -1);
    if (forced > 0) { w.fold = 0; return grid; }          // (a forced group is taken as it is: no rounding up to whole GQA groups)
    const int hpx = nheads >> 3;
    if (gqa_group > 1 && hpx > 0) {
        int g = ((w.group + gqa_group - 1) / gqa_group) * gqa_group;
        g = g > hpx ? hpx : g;
        if (g != w.group) w.fold = 0;              // the fold needs one sorted list per XCD (plan_work_order decided it for its own group)
        w.group = g;
        const int cnt = w.left * ((nitems + 7) / 8) + hpx * nitems;
        const bool one_sorted_list = (w.left == 0 && w.group >= hpx);
        w.fold = ((head_dim == 64 ? 3 : 2) == 2 && cnt > 32 && cnt <= 64 && one_sorted_list) ? 1 : 0;
    }
    return grid;
}

}  // namespace sage
