// sage_attn_d64_f8.hip -- instantiation unit of the attention kernel family (sage_attn_kernel.h): launch_attn_part<D, PV_FP8, SFOLD> = <64,true,false>
// (FP8 PV, the exact score form: the default of every FP8 entry point)
#include "sage_attn_launch.h"
namespace sage {
template hipError_t launch_attn_part<64,true,false>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
}
