// sage_attn_d128_f8.hip -- instantiation unit of the attention kernel family (sage_attn_kernel.h): launch_attn_part<D, PV_FP8, SFOLD> = <128,true,false>
// (FP8 PV, the exact score form: the default of every FP8 entry point)
#include "sage_attn_launch.h"
namespace sage {
template hipError_t launch_attn_part<128,true,false>(const AttnParams &, const AttnVariant &, int, const AttnLaunchOpts &);
}
