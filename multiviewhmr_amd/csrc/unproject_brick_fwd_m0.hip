// k_fwd_brick instantiations for aggregation method 0 (one translation unit per method: parallel builds)
#include "brick_fwd_ws.h"
namespace mvhmr {
template hipError_t launch_fwd_method<0>(const void *, const float *, const Coords &, void *, const Problem &, int, hipStream_t);
}
#if MVHMR_EXP & 1024
// experiment builds only (scripts/exp): phase timers of k_fwd_brick
extern "C" __attribute__((visibility("default"))) int mvhmr_exp_fwd_timers_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mvhmr::g_exp_fwd_timers), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mvhmr::g_exp_fwd_timers), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
