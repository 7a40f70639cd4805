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

// SelfNorm (one gate; optional PRE add / ReLU epilogue), alone or behind CrossNorm without crop boxes and without the
// channel permutation, on planes of at most 64 elements that are not a whole number of 8-byte vectors (7x7, 5x5, 3x3),
// C a multiple of 16 / element bytes, N <= 256.  perm: the batch permutation (device, int64[N]) when CrossNorm is active.
WidePlan wide_plan(const Plan& pl, int add, bool backward, bool has_chan_perm = false);

int wide_forward(const Plan& pl, const WidePlan& wp, int add, int relu, const void* x, const void* addend, const int64_t* perm,
                 GateDev g, void* y, double* saved, hipStream_t stream);
int wide_backward(const Plan& pl, const WidePlan& wp, int add, int relu, const void* gy, const void* x, const void* addend,
                  const int64_t* perm, GateDev g, const double* saved, void* dx, GateGradDev dg, hipStream_t stream);

}  // namespace cnsn
