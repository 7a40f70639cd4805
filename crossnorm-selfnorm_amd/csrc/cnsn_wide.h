// Channel-group-in-registers strategy (cnsn_wide_kernels.h): host entry points.
#pragma once
#include "cnsn_host_plan.h"

namespace cnsn {

struct WidePlan {
    bool ok;
    int vec;  // elements per lane = channels per workgroup
    int R;    // instances per wave
    size_t lds;
};

// SelfNorm alone (one gate; optional PRE add / ReLU epilogue) on planes of at most 64 elements that are not a whole
// number of 8-byte vectors (7x7, 5x5, 3x3), C a multiple of 16 / element bytes, N <= 256
WidePlan wide_plan(const Plan& pl, int add, bool backward);

int wide_forward(const Plan& pl, const WidePlan& wp, int add, int relu, const void* x, const void* addend, GateDev g, void* y,
                 double* saved, hipStream_t stream);
int wide_backward(const Plan& pl, const WidePlan& wp, int add, int relu, const void* gy, const void* x, const void* addend,
                  GateDev g, const double* saved, void* dx, GateGradDev dg, hipStream_t stream);

}  // namespace cnsn
