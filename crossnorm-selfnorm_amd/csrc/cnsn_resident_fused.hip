// Channel-resident strategy with the residual-block epilogue: host side.
#include "cnsn_resident_fused.h"

namespace cnsn {

ResPlan resident_fused_plan(const cnsn_problem_t&, bool, bool, int, bool) { return ResPlan{false, 0, 0, 0, 0}; }

int resident_fused_forward(const cnsn_problem_t&, Box, Box, bool, const MidArgs&, int, int, const void*, const void*,
                           const int64_t*, GateDev, GateDev, void*, double*, void*, hipStream_t) {
    return CNSN_E_UNSUPPORTED;
}
int resident_fused_backward(const cnsn_problem_t&, Box, Box, bool, const MidArgs&, int, int, const void*, const void*,
                            const void*, const int64_t*, GateDev, GateDev, const double*, void*, GateGradDev, GateGradDev,
                            void*, hipStream_t) {
    return CNSN_E_UNSUPPORTED;
}

}  // namespace cnsn
