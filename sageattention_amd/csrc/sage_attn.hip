// sage_attn.hip -- host-side dispatch of the attention launches: the work order of a launch (sage_work_order.h) and the choice of the
// instantiation unit (sage_attn_parts.h; the kernels themselves are in sage_attn_kernel.h, compiled by sage_attn_d*_*.hip).
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_attn_parts.h"
#include "sage_work_order.h"
#include <cstdlib>

#ifndef SAGE_ORDER_DEFAULT   // causal work order: -1 = grouped / folded (set_work_order), 0 = head-major heavy-first, n = groups of n heads
#define SAGE_ORDER_DEFAULT -1
#endif

namespace sage {

// Causal dense grids: which (head, query block) item a workgroup takes -- sage_work_order.h (mapping, group-size rule, measurements).
// Split-KV chunks (weights depend on the chunk), masked and varlen calls keep the head-major heavy-first order over contiguous runs.
// SAGE_ORDER_GROUP / sage_set_work_order: 0 restores the head-major order, n > 0 forces the group size (experiments).
static int g_work_order = -2;         // -2: not read yet
int work_order()
{
    if (g_work_order == -2) {
        const char *e = getenv("SAGE_ORDER_GROUP");
        g_work_order = (e != nullptr && e[0] != 0) ? atoi(e) : SAGE_ORDER_DEFAULT;
    }
    return g_work_order;
}
void set_work_order_mode(int group) { g_work_order = group; }

static int set_work_order(AttnParams &q, bool causal, int head_dim, bool pv_fp8, bool masked)
{
    q.order_group = 0;
    q.order_fold = 0;
    q.order_left = 0;
    const int nheads = q.B * q.Hq;
    if (q.cu_q != nullptr && q.work_items != nullptr)             // varlen, device-built work list: bound of the dense-style grid over it
        return 8 * ((q.Hq & 7) * ((q.items_bound + 7) / 8) + (q.Hq >> 3) * q.items_bound);
    if (q.cu_q != nullptr) return ((q.B * q.Hkv + 7) / 8) * 8 * q.group * q.nqblk;   // varlen: whole rounds of 8 (sequence, kv-head) units
    const int forced = work_order();
    if (!causal || masked || q.kv_split > 1 || q.nqblk <= 1 || forced == 0) return nheads * q.nqblk;
    WorkOrder w;
    const int grid = plan_work_order(w, nheads, q.nqblk, q.Lk, head_dim, pv_fp8, forced);
    q.order_group = w.group;
    q.order_fold = w.fold;
    q.order_left = w.left;
    return grid;
}

// the instantiation unit of (head_dim, PV format, FP8 score form)
static hipError_t launch_unit(const AttnParams &p, int head_dim, bool pv_fp8, const AttnVariant &v, int nwork, const AttnLaunchOpts &o)
{
    if (head_dim == 128) {
        if (!pv_fp8) return launch_attn_part<128, false, true>(p, v, nwork, o);
        return o.fp8_folded ? launch_attn_part<128, true, true>(p, v, nwork, o) : launch_attn_part<128, true, false>(p, v, nwork, o);
    }
    if (head_dim == 64) {
        if (!pv_fp8) return launch_attn_part<64, false, true>(p, v, nwork, o);
        return o.fp8_folded ? launch_attn_part<64, true, true>(p, v, nwork, o) : launch_attn_part<64, true, false>(p, v, nwork, o);
    }
    return hipErrorInvalidValue;
}

// per-thread granularity, q in fp16 (q_dtype 0) / bf16 (1), quantised in the kernel prologue
hipError_t launch_attn_fused_q(const AttnParams &p_in, int head_dim, bool causal, int q_dtype, bool pv_fp8, const AttnLaunchOpts &o)
{
    AttnParams p = p_in;
    const int nwork = set_work_order(p, causal, head_dim, pv_fp8, false);
    if (o.grid_out != nullptr) *o.grid_out = 0;
    if (nwork <= 0) return hipSuccess;
    if (p.cu_q != nullptr || (q_dtype != DT_F16 && q_dtype != DT_BF16)) return hipErrorInvalidValue;
    if (p.v_rows != 0 && (pv_fp8 || q_dtype != DT_F16 || p.kv_split > 1)) return hipErrorInvalidValue;
    const AttnVariant v = {causal, true, pv_fp8, 0, q_dtype == DT_F16 ? 1 : 2, p.v_rows != 0};
    return launch_unit(p, head_dim, pv_fp8, v, nwork, o);
}

hipError_t launch_attn_fused_qblock(const AttnParams &p_in, int head_dim, bool causal, int q_dtype, const AttnLaunchOpts &o)
{
    AttnParams p = p_in;
    const int nwork = set_work_order(p, causal, head_dim, false, false);
    if (o.grid_out != nullptr) *o.grid_out = 0;
    if (nwork <= 0) return hipSuccess;
    if (q_dtype != DT_F16 && q_dtype != DT_BF16) return hipErrorInvalidValue;
    if (p.v_rows != 0 && (q_dtype != DT_F16 || p.cu_q != nullptr)) return hipErrorInvalidValue;
    const AttnVariant v = {causal, false, true, 0, q_dtype == DT_F16 ? 3 : 4, p.v_rows != 0};
    return launch_unit(p, head_dim, false, v, nwork, o);
}

hipError_t launch_attn(const AttnParams &p_in, int head_dim, bool pv_fp8, bool causal, bool kthread,
                       bool two_level, int mask_kind, const AttnLaunchOpts &o)
{
    AttnParams p = p_in;
    const int nwork = set_work_order(p, causal, head_dim, pv_fp8, mask_kind != 0);
    if (o.grid_out != nullptr) *o.grid_out = 0;
    if (nwork <= 0) return hipSuccess;
    if (mask_kind != 0 && (pv_fp8 || causal || kthread || mask_kind < 1 || mask_kind > 3)) return hipErrorInvalidValue;
    if (p.v_rows != 0 && (pv_fp8 || mask_kind != 0 || p.cu_q != nullptr || p.kv_split > 1)) return hipErrorInvalidValue;
    const AttnVariant v = {causal, kthread, mask_kind != 0 ? true : two_level, mask_kind, 0, p.v_rows != 0};
    return launch_unit(p, head_dim, pv_fp8, v, nwork, o);
}

}  // namespace sage
