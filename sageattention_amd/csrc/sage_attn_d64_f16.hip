// sage_attn_d64_f16.hip -- instantiation unit of the attention kernel family (sage_attn_kernel.h): launch_attn_part<D, PV_FP8, SFOLD> = <64,false,true>
#include "sage_attn_launch.h"
namespace sage {
template hipError_t launch_attn_part<64,false,true>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
}
