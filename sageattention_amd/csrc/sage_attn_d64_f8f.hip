// sage_attn_d64_f8f.hip -- instantiation unit of the attention kernel family (sage_attn_kernel.h): launch_attn_part<D, PV_FP8, SFOLD> = <64,true,true>
// (FP8 PV, the folded score form: the opt-in variant SAGE_ATTR_FP8_FOLDED_SCORES)
#include "sage_attn_launch.h"
namespace sage {
template hipError_t launch_attn_part<64,true,true>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
}
