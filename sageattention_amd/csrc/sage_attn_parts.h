// sage_attn_parts.h -- interface between the host-side dispatch of the attention launches (sage_attn.hip) and the instantiation units
// sage_attn_d{128,64}_{f8,f8f,f16}.hip, each of which compiles the kernel family of sage_attn_kernel.h for one head size, one PV format and
// (FP8) one score form.  The split exists for build time only: the units are independent and compile in parallel.
#pragma once
#include "sage_kernels.h"

namespace sage {

// which member of the kernel family a launch takes (everything the template arguments of sage_attn_kernel encode besides the unit's own)
struct AttnVariant {
    bool causal;
    bool kthread;       // per-thread k scale groups (4 per 64 keys)
    bool two_level;     // FP8 PV: tile product from a zero accumulator; FP16 PV: the Triton kernel form (SAGE_PV_ACCUM_TRITON)
    int mask_kind;      // 0 none, 1 bool, 2 additive fp16, 3 additive bf16 (FP16 PV, per-block scales, non-causal)
    int qf;             // 0: INT8 q + q_scale; 1 / 2: fp16 / bf16 q quantised per thread group in the prologue; 3 / 4: per 128-row block
    bool vrows;         // FP16 PV, qf 0 / 1 / 3, dense, unmasked: AttnParams::v is the caller's fp16 V (rows, v_sb / v_sh / v_sl), not the tile image
};

// D in {64, 128}; PV_FP8; SFOLD: the FP8 score form (true = folded bias, false = exact subtraction; FP16 PV has one form: true)
template <int D, bool PV_FP8, bool SFOLD>
hipError_t launch_attn_part(const AttnParams &p, const AttnVariant &v, int nwork, const AttnLaunchOpts &l);

extern template hipError_t launch_attn_part<128, true, true>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
extern template hipError_t launch_attn_part<128, true, false>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
extern template hipError_t launch_attn_part<128, false, true>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
extern template hipError_t launch_attn_part<64, true, true>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
extern template hipError_t launch_attn_part<64, true, false>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
extern template hipError_t launch_attn_part<64, false, true>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);

}  // namespace sage
