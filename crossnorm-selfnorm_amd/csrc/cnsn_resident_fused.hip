// Channel-resident strategy with the residual-block epilogue (add before the op, ReLU after): host entry points.
// The POST add mode (addend joins after the op) is offered for un-boxed calls: its planes are fetched after the cluster
// exchange instead of being held across it; boxed POST calls run the two-pass kernels.
#include "cnsn_resident_fused.h"

#include "cnsn_fused_stream_kernels.h"
#include "cnsn_resident_host.h"

namespace cnsn {

ResPlan resident_fused_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int add, bool backward) {
    if (add == ADD_POST && boxed) return ResPlan{false, 0, 0, 0, 0};
    return reshost::plan_impl(p, boxed, has_chan_perm, backward, true, add == ADD_POST);
}

int resident_fused_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                           const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                           double* saved, void* workspace, hipStream_t stream) {
    if (add == ADD_POST && boxed) return CNSN_E_UNSUPPORTED;
    return reshost::forward_impl<true>(p, cb, sb, boxed, mid, x, add != ADD_NONE ? addend : nullptr, relu, perm, g, f, y,
                                       saved, workspace, stream, add == ADD_POST);
}

}  // namespace cnsn
