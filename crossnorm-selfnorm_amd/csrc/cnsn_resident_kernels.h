// Channel-resident strategy (one launch per direction, the channel's planes stay on chip).
#pragma once
#include "../../include/cnsn_hip.h"
#include "cnsn_device.h"
#include "cnsn_mid_kernels.h"

namespace cnsn {

inline bool use_resident(const cnsn_problem_t& p, bool boxed, bool has_chan_perm) {
    (void)p; (void)boxed; (void)has_chan_perm;
    return false;
}

inline int resident_forward(const cnsn_problem_t&, Box, Box, bool, const MidArgs&, const void*, const int64_t*,
                            GateDev, GateDev, void*, double*, float*, hipStream_t) {
    return CNSN_E_UNSUPPORTED;
}
inline int resident_backward(const cnsn_problem_t&, Box, Box, bool, const MidArgs&, const void*, const void*,
                             const int64_t*, GateDev, GateDev, const double*, void*, GateGradDev, GateGradDev, float*,
                             hipStream_t) {
    return CNSN_E_UNSUPPORTED;
}

}  // namespace cnsn
