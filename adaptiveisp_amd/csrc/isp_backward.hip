// Parameter-gradient kernels (placeholder until the forward path is verified on hardware).
#include "isp_internal.h"
namespace adaisp {
hipError_t launch_backward_params(const float*, const float*, const int32_t*, const float*, int, float*, int, int, int,
                                  unsigned, hipStream_t) {
    return hipErrorNotSupported;
}
}  // namespace adaisp
