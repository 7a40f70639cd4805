// Channels-last two-pass path: host entry points (cnsn_nhwc.hip).  `cnsn_problem_t.layout` = CNSN_LAYOUT_NHWC.
#pragma once
#include "cnsn_host_plan.h"

namespace cnsn {

// the un-boxed op on a channels-last tensor whose channel count is a whole number of 16-byte vectors, no channel permutation
bool nhwc_supported(const Plan& pl, bool has_chan_perm);
// bytes the channels-last path needs behind the two-pass workspace (partial sums of the pixel chunks, plane-order rows)
size_t nhwc_extra_bytes(const Plan& pl);
int nhwc_forward(Plan& pl, int add, int relu, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                 float* saved, void* workspace, size_t workspace_bytes, hipStream_t stream);
int nhwc_backward(Plan& pl, int add, int relu, const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g,
                  GateDev f, const float* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                  size_t workspace_bytes, hipStream_t stream);

}  // namespace cnsn
