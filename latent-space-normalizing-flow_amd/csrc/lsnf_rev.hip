// placeholder until the reverse kernel lands (fails loudly, no fallback)
#include <hip/hip_runtime.h>
#include "lsnf_layout.h"
hipError_t lsnf_launch_reverse(const LsnfGeo&, const float*, int, const float*, const float*, float*, float*, int, hipStream_t) {
    return hipErrorNotSupported;
}
