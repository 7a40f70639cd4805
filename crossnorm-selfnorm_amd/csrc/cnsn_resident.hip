// Channel-resident strategy, the op alone: host entry points (shared logic in cnsn_resident_host.h).
#include "cnsn_resident_host.h"

namespace cnsn {

ResPlan resident_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, bool backward) {
    return reshost::plan_impl(p, boxed, has_chan_perm, backward, false);
}

int resident_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* x,
                     const int64_t* perm, GateDev g, GateDev f, void* y, double* saved, void* workspace,
                     hipStream_t stream) {
    return reshost::forward_impl<false>(p, cb, sb, boxed, mid, x, nullptr, 0, perm, g, f, y, saved, workspace, stream);
}

int resident_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy,
                      const void* x, const int64_t* perm, GateDev g, GateDev f, const double* saved, void* dx,
                      GateGradDev dg, GateGradDev df, void* workspace, hipStream_t stream) {
    return reshost::backward_impl<false>(p, cb, sb, boxed, mid, gy, x, nullptr, 0, perm, g, f, saved, dx, dg, df,
                                         workspace, stream);
}

size_t resident_workspace_bytes(const cnsn_problem_t& p, bool boxed) {
    return kCtlBytes + (size_t)p.N * p.C * (boxed ? 6 : 2) * 8;
}

}  // namespace cnsn
