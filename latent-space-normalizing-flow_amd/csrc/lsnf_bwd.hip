// placeholder until the backward kernel lands (fails loudly, no fallback)
#include <hip/hip_runtime.h>
#include "lsnf_layout.h"
hipError_t lsnf_launch_backward_z(const LsnfGeo&, const float*, int, const float*, const float*, const float*, const float*, int, float, float*, int, hipStream_t) {
    return hipErrorNotSupported;
}
