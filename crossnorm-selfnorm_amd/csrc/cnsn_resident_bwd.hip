// Channel-resident strategy, the op alone, BACKWARD: a translation unit of its own (round 5: compiled in parallel with the forward).
#include "cnsn_resident_host.h"
#include "cnsn_env.h"

namespace cnsn {

int resident_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy,
                      const void* x, const int64_t* perm, GateDev g, GateDev f, const double* saved, void* dx,
                      GateGradDev dg, GateGradDev df, void* workspace, hipStream_t stream) {
    return reshost::backward_impl<false>(p, cb, sb, boxed, mid, gy, x, nullptr, 0, perm, g, f, saved, dx, dg, df,
                                         workspace, stream);
}

}  // namespace cnsn
