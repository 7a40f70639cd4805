// Channel-resident strategy with the residual-block epilogue, BACKWARD: a translation unit of its own (round 5: the forward
// and backward instantiations compile in parallel — this family alone was 2 m 20 s of a cold build).
#include "cnsn_resident_fused.h"

#include "cnsn_fused_stream_kernels.h"
#include "cnsn_resident_host.h"

namespace cnsn {

int resident_fused_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                            const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f,
                            const double* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                            hipStream_t stream) {
    if (add == ADD_POST && (boxed || !relu || !d_addend)) return CNSN_E_UNSUPPORTED;
    return reshost::backward_impl<true>(p, cb, sb, boxed, mid, gy, x, add != ADD_NONE ? addend : nullptr, relu, perm, g, f,
                                        saved, dx, dg, df, workspace, stream, add == ADD_POST, d_addend);
}

}  // namespace cnsn
