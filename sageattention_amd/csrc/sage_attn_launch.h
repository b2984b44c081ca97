// sage_attn_launch.h -- host side of the attention kernel family: the one launch path every instantiation takes (dynamic-LDS opt-in, the
// persistent / ticket route) and the map from a launch's variant to the instantiation of sage_attn_kernel.h that runs it.  Included by the
// instantiation units sage_attn_d{128,64}_{f8,f8f,f16}.hip behind the kernel header.
#pragma once
#include "sage_attn_kernel.h"
#include <atomic>

namespace sage {

// ---------------------------------------------------------------------------------------------
// One launch path for every instantiation.  KERN is a non-type template argument, so the statics below exist once per kernel: the
// > 64 KiB dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is issued once per kernel and device, and the number of
// workgroups the device holds of THIS kernel is probed once per kernel and device (both lock-free: a race repeats the idempotent probe).
constexpr int kPersistMinRounds = 12;

template <auto KERN>
static hipError_t launch_kernel(int lds, const AttnParams &p, int nwork, const AttnLaunchOpts &l, bool persist_ok)
{
    static std::atomic<unsigned long long> lds_done{0};      // bit per device ordinal (mod 64)
    static std::atomic<int> slots_of[64];                    // per device ordinal (mod 64): 0 = not probed yet, else 1 + resident workgroups
    int dev = -1;
    const bool dev_ok = hipGetDevice(&dev) == hipSuccess && dev >= 0;
    if (lds > 65536) {
        if (!dev_ok) return hipErrorInvalidDevice;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(lds_done.load(std::memory_order_relaxed) & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            lds_done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    AttnParams pp = p;
    int grid = nwork;
    // Persistent launch (AttnParams::sched, a zeroed counter block of the caller; non-causal unmasked kernels): as many workgroups as the
    // device holds at once take the logical workgroup indices 0 .. nwork - 1 from 32 ticket queues (see the kernel).  Worth it from twelve
    // rounds of workgroups up -- one item must be small against the few per cent the XCDs differ by, or nothing can be evened out: with
    // eight rounds of equal items (B2 H32 N8192 non-causal) the tickets cost 1 % -- anything else is an ordinary launch.
    // What the route NEEDS (checked here, else the ordinary launch): the grid a multiple of 32 (the first round then covers whole tickets of
    // every queue) and not larger than the logical grid (every first-round workgroup has an item).  What it merely ASSUMES, for speed only:
    // blockIdx.x & 7 = the XCD of a first-round workgroup, and that the grid is co-resident (an unpartitioned 256-CU device, no CU mask).  On a
    // partitioned device or a CU-masked stream every logical index is still taken exactly once (tests/test_work_order.py restates the
    // partition); the queues then no longer match the L2s.  A failing occupancy probe is not an error of the call: ordinary launch.
    if (pp.sched != nullptr) {
        unsigned *sched = pp.sched;
        pp.sched = nullptr;
        if (persist_ok && dev_ok && (nwork & 7) == 0) {
            int slots = slots_of[dev & 63].load(std::memory_order_relaxed) - 1;
            if (slots < 0) {
                int ncu = 0, per_cu = 0;
                if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(KERN), 256, lds) == hipSuccess)
                    slots = ncu * per_cu;
                else {
                    slots = 0;
                    (void)hipGetLastError();
                }
                slots_of[dev & 63].store(slots + 1, std::memory_order_relaxed);
            }
            const bool rounds_ok = l.force_persistent ? nwork >= 2 * slots : nwork >= kPersistMinRounds * slots;
            if (slots > 0 && (slots & 31) == 0 && rounds_ok) { pp.sched = sched; pp.nwg = nwork; grid = slots; }
        }
    }
    if (l.grid_out != nullptr) *l.grid_out = grid;
    hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), lds, l.stream, pp);
    return hipGetLastError();
}

// The members of the family one instantiation unit holds: INT8 q (8 of causal x k-scale groups x accumulation), the fused per-thread Q
// quantiser (fp16 / bf16 q), and for FP16 PV the masked kernels and the fused per-block Q quantiser.
// Keys per iteration: 64 (NH = 1).
template <int D, bool PV_FP8, bool SFOLD>
hipError_t launch_attn_part(const AttnParams &p, const AttnVariant &v, int nwork, const AttnLaunchOpts &l)
{
    static_assert(PV_FP8 || SFOLD, "FP16 PV has one score form");
    constexpr int NH = 1;
    using C = TileCfg<D, PV_FP8, NH>;
    // causal launches take the ticket route only over a packed batch's work list (items of very different lengths, heaviest first: +2.9 % at
    // C4), in instantiations of their own (CPERS); dense causal launches keep the hardware's dispatch, which their work order is built on
    const bool packed_list = p.cu_q != nullptr && p.work_items != nullptr;
    const bool pers = !v.causal;
    if (v.mask_kind != 0) {       // Triton-named API: FP16 PV, per-block scales, non-causal, tile product folded into the FP32 output
        if constexpr (!PV_FP8) {
            using CM = TileCfg<D, false, 1>;
            if (v.causal || v.kthread || v.qf != 0) return hipErrorInvalidValue;
            if (v.mask_kind == 1) return launch_kernel<sage_attn_kernel<D, false, false, false, true, 1, 1>>(CM::LDS_BYTES, p, nwork, l, false);
            if (v.mask_kind == 2) return launch_kernel<sage_attn_kernel<D, false, false, false, true, 1, 2>>(CM::LDS_BYTES, p, nwork, l, false);
            if (v.mask_kind == 3) return launch_kernel<sage_attn_kernel<D, false, false, false, true, 1, 3>>(CM::LDS_BYTES, p, nwork, l, false);
        }
        return hipErrorInvalidValue;
    }
    if (v.vrows) {                     // V rows read in place (fp16 V, FP16 PV, dense): fused Q quantisation per thread group / per block, or INT8 q
        if constexpr (!PV_FP8) {
            if (packed_list || p.cu_q != nullptr) return hipErrorInvalidValue;
#define SAGE_VR(C_) \
            if (v.causal == C_ && v.qf == 1) return launch_kernel<sage_attn_kernel<D, false, C_, true, false, NH, 0, 1, true, false, true>>(C::LDS_BYTES, p, nwork, l, pers); \
            if (v.causal == C_ && v.qf == 3) return launch_kernel<sage_attn_kernel<D, false, C_, false, true, NH, 0, 3, true, false, true>>(C::LDS_BYTES, p, nwork, l, pers);
            SAGE_VR(false) SAGE_VR(true)
#undef SAGE_VR
            // INT8 q with its scales (ABI 21, sage_attn_qk_int8_pv_f16_vrows): the reference's native FP16-PV ops as they are -- query / key INT8,
            // value fp16 rows (pybind_sm80.cpp:21-27; the Triton forward, attn_qk_int8_per_block.py:130) -- in every k-scale grouping and both kernel forms
#define SAGE_VR0(C_, K_, T_) if (v.qf == 0 && v.causal == C_ && v.kthread == K_ && v.two_level == T_) \
            return launch_kernel<sage_attn_kernel<D, false, C_, K_, T_, NH, 0, 0, true, false, true>>(C::LDS_BYTES, p, nwork, l, pers);
            SAGE_VR0(false, false, false) SAGE_VR0(false, false, true) SAGE_VR0(true, false, false) SAGE_VR0(true, false, true)
            SAGE_VR0(false, true, false)  SAGE_VR0(false, true, true)  SAGE_VR0(true, true, false)  SAGE_VR0(true, true, true)
#undef SAGE_VR0
        }
        return hipErrorInvalidValue;
    }
    if (v.qf == 1 || v.qf == 2) {      // per-thread groups quantised in the prologue; FP8 PV: two-level, FP16 PV: straight FP32 accumulation
#define SAGE_FQ(C_, F_) if (v.causal == C_ && v.qf == F_) return launch_kernel<sage_attn_kernel<D, PV_FP8, C_, true, PV_FP8, NH, 0, F_, SFOLD>>(C::LDS_BYTES, p, nwork, l, pers);
        SAGE_FQ(false, 1) SAGE_FQ(false, 2) SAGE_FQ(true, 1) SAGE_FQ(true, 2)
#undef SAGE_FQ
        return hipErrorInvalidValue;
    }
    if (v.qf == 3 || v.qf == 4) {      // per-block Q in the prologue: the Triton-named API's kernels (FP16 PV, per-block k scales), dense or varlen
        if constexpr (!PV_FP8) {
            if (v.causal && packed_list) {
                if (v.qf == 3) return launch_kernel<sage_attn_kernel<D, false, true, false, true, NH, 0, 3, true, true>>(C::LDS_BYTES, p, nwork, l, true);
                return launch_kernel<sage_attn_kernel<D, false, true, false, true, NH, 0, 4, true, true>>(C::LDS_BYTES, p, nwork, l, true);
            }
#define SAGE_FQB(C_, F_) if (v.causal == C_ && v.qf == F_) return launch_kernel<sage_attn_kernel<D, false, C_, false, true, NH, 0, F_>>(C::LDS_BYTES, p, nwork, l, pers);
            SAGE_FQB(false, 3) SAGE_FQB(false, 4) SAGE_FQB(true, 3) SAGE_FQB(true, 4)
#undef SAGE_FQB
        }
        return hipErrorInvalidValue;
    }
#define SAGE_CASE(C_, K_, T_) if (v.causal == C_ && v.kthread == K_ && v.two_level == T_) \
        return launch_kernel<sage_attn_kernel<D, PV_FP8, C_, K_, T_, NH, 0, 0, SFOLD>>(C::LDS_BYTES, p, nwork, l, pers);
    SAGE_CASE(false, false, false) SAGE_CASE(false, false, true)
    SAGE_CASE(true, false, false)  SAGE_CASE(true, false, true)
    SAGE_CASE(false, true, false)  SAGE_CASE(false, true, true)
    SAGE_CASE(true, true, false)   SAGE_CASE(true, true, true)
#undef SAGE_CASE
    return hipErrorInvalidValue;
}

}  // namespace sage
