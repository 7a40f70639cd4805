// Channel-in-registers strategy (cnsn_mono_kernels.h): host entry points.
#pragma once
#include "cnsn_host_plan.h"

namespace cnsn {

struct MonoPlan {
    bool ok;
    int vec, lpp, rmax, R;
    size_t lds;
};

// SelfNorm alone (no CrossNorm; optional PRE add / ReLU epilogue) on planes of at most 64 vectors of 8 or 16 bytes,
// a whole channel in the registers of one 1024-thread workgroup
MonoPlan mono_plan(const Plan& pl, int add, bool backward);

int mono_forward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* x, const void* addend, GateDev g,
                 GateDev f, void* y, double* saved, hipStream_t stream);
int mono_backward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* gy, const void* x, const void* addend,
                  GateDev g, GateDev f, const double* saved, void* dx, GateGradDev dg, GateGradDev df, hipStream_t stream);

// the same frame with CrossNorm (cnsn_mono_cn_kernels.h): cn_active, any boxes / lam, optional SelfNorm (one gate),
// optional PRE add / ReLU; no channel permutation
MonoPlan mono_cn_plan(const Plan& pl, bool has_chan_perm, int add, bool backward);
int mono_cn_forward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* x, const void* addend,
                    const int64_t* perm, GateDev g, void* y, double* saved, hipStream_t stream);
int mono_cn_backward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* gy, const void* x, const void* addend,
                     const int64_t* perm, GateDev g, const double* saved, void* dx, GateGradDev dg, hipStream_t stream);

}  // namespace cnsn
