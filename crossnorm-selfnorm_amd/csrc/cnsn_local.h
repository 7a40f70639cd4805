// Channel-local strategy (cnsn_local_kernels.h): host entry points.
#pragma once
#include "cnsn_host_plan.h"

namespace cnsn {

struct LocalPlan {
    bool ok;
    int CG, W;
    size_t lds;
    int block;  // threads per workgroup
};

// SelfNorm alone (no CrossNorm) on planes small enough that a whole channel (group) fits one workgroup's LDS
LocalPlan local_plan(const Plan& pl, int add, bool backward);

int local_forward(const Plan& pl, const LocalPlan& lp, int add, int relu, const void* x, const void* addend, GateDev g,
                  GateDev f, void* y, double* saved, hipStream_t stream);
int local_backward(const Plan& pl, const LocalPlan& lp, int add, int relu, const void* gy, const void* x,
                   const void* addend, GateDev g, GateDev f, const double* saved, void* dx, GateGradDev dg, GateGradDev df,
                   hipStream_t stream);

}  // namespace cnsn
