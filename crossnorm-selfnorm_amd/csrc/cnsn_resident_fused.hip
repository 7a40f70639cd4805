// Channel-resident strategy with the residual-block epilogue (add before the op, ReLU after): host entry points.
// The POST add mode is not offered by the resident kernels (the addend would have to stay in registers across the
// cluster exchange): such calls run the two-pass kernels.
#include "cnsn_resident_fused.h"

#include "cnsn_fused_stream_kernels.h"
#include "cnsn_resident_host.h"

namespace cnsn {

ResPlan resident_fused_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int add, bool backward) {
    if (add == ADD_POST) return ResPlan{false, 0, 0, 0, 0};
    return reshost::plan_impl(p, boxed, has_chan_perm, backward, true);
}

int resident_fused_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                           const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                           double* saved, void* workspace, hipStream_t stream) {
    if (add == ADD_POST) return CNSN_E_UNSUPPORTED;
    return reshost::forward_impl<true>(p, cb, sb, boxed, mid, x, add == ADD_PRE ? addend : nullptr, relu, perm, g, f, y,
                                       saved, workspace, stream);
}

int resident_fused_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                            const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f,
                            const double* saved, void* dx, GateGradDev dg, GateGradDev df, void* workspace,
                            hipStream_t stream) {
    if (add == ADD_POST) return CNSN_E_UNSUPPORTED;
    return reshost::backward_impl<true>(p, cb, sb, boxed, mid, gy, x, add == ADD_PRE ? addend : nullptr, relu, perm, g, f,
                                        saved, dx, dg, df, workspace, stream);
}

}  // namespace cnsn
