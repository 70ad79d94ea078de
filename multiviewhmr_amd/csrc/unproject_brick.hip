// "brick" variant: voxel bricks with the feature patches they touch staged in LDS.  (placeholder until the
// kernel lands; the dispatcher falls back to the gather variant while brick_supported() is false.)
#include "device_common.h"
#include "kernels.h"

namespace mvhmr {

bool brick_supported(const Problem &) { return false; }
size_t brick_workspace_bytes(const Problem &) { return 0; }
hipError_t launch_fwd_brick(const void *, bool, const float *, const float *, void *, void *, const Problem &, hipStream_t)
{
    return hipErrorNotSupported;
}

}  // namespace mvhmr
