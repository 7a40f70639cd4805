// Channel-resident kernels with the residual-block epilogue (host entry points; kernels in
// cnsn_resident_fused_kernels.h, built by cnsn_resident_fused.hip).
#pragma once
#include "cnsn_host_plan.h"
#include "cnsn_resident_kernels.h"

namespace cnsn {

ResPlan resident_fused_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int add, bool backward);

int resident_fused_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                           const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                           double* saved, void* workspace, hipStream_t stream);
int resident_fused_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                            const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f,
                            const double* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                            hipStream_t stream);

}  // namespace cnsn
