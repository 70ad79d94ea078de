// k_fwd_brick instantiations for aggregation method 0 (one translation unit per method: parallel builds)
#include "brick_fwd_ws.h"
namespace mvhmr {
template hipError_t launch_fwd_method<0>(const void *, const float *, const Coords &, void *, const Problem &, int, hipStream_t);
}
