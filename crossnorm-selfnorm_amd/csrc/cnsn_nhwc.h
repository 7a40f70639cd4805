// Channels-last two-pass path: host entry points (cnsn_nhwc.hip).  `cnsn_problem_t.layout` = CNSN_LAYOUT_NHWC.
#pragma once
#include "cnsn_host_plan.h"

namespace cnsn {

// the un-boxed op on a channels-last tensor whose channel count is a whole number of 16-byte vectors, no channel permutation
bool nhwc_supported(const Plan& pl, bool has_chan_perm);
// bytes the channels-last path needs behind the two-pass workspace (partial sums of the pixel chunks, plane-order rows)
size_t nhwc_extra_bytes(const Plan& pl);
// sum_out (ADD_PRE only, may be null, may alias x): where X = x + addend is kept (cnsn_epilogue_t.sum_out, ABI 8)
int nhwc_forward(Plan& pl, int add, int relu, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                 float* saved, void* workspace, size_t workspace_bytes, hipStream_t stream, void* sum_out = nullptr);
int nhwc_backward(Plan& pl, int add, int relu, const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g,
                  GateDev f, const float* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                  size_t workspace_bytes, hipStream_t stream);

// whole workspace a channels-last call may need (either strategy)
size_t nhwc_workspace_bytes(const Plan& pl);

// ---- single-launch kernels (cnsn_nhwc_fused.hip; SelfNorm alone, one gate, N <= 256) and the slim `saved` record they share with
// the two-pass kernels (cnsn_nhwc_fused_kernels.h)
bool nhwc_slim_record(const Plan& pl);  // this call's `saved` is the slim record (channels-last, no CrossNorm, one gate)
bool nhwc_fused_ok(const Plan& pl, bool check_health = true);  // the single-launch kernels take the call (strategy, switches, health, shape)
size_t nhwc_fused_extra_bytes(const Plan& pl);
int nhwc_fused_forward(Plan& pl, int add, int relu, const void* x, const void* addend, GateDev g, void* y, float* saved,
                       void* workspace, size_t workspace_bytes, hipStream_t stream, void* sum_out = nullptr);
int nhwc_fused_backward(Plan& pl, int add, int relu, const void* gy, const void* x, const void* addend, GateDev g,
                        const float* saved, void* dx, void* d_addend, GateGradDev dg, void* workspace, size_t workspace_bytes,
                        hipStream_t stream);
void nhwc_slim_from_saved(const Plan& pl, const double* saved_d, float* slim, hipStream_t stream);
void nhwc_saved_from_slim(const Plan& pl, const float* slim, int relu, double* saved_d, float* rows, hipStream_t stream);

// ---- the block's last BatchNorm2d in front of the op (cnsn_nhwc_bnhead_kernels.h; training mode, SelfNorm alone, one gate, N <= 256)
bool nhwc_bnhead_ok(const Plan& pl, bool check_health = true);
size_t nhwc_bnhead_extra_bytes(const Plan& pl);
// bn2 (may be null): the skip path ends in a BatchNorm2d of its own — `identity` is then ITS input (the downsample convolution's output)
int nhwc_bnhead_forward(Plan& pl, int relu, const cnsn_bn_tail_t& bn, const cnsn_bn_tail_t* bn2, const void* conv_out, const void* identity,
                        GateDev gg, void* y, float* saved, float* bn_stats, void* workspace, size_t workspace_bytes, hipStream_t stream);
int nhwc_bnhead_backward(Plan& pl, int relu, const cnsn_bn_tail_t& bn, const cnsn_bn_tail_t* bn2, const void* gy, const void* conv_out,
                         const void* identity, GateDev gg, const float* saved, const float* bn_stats, void* d_conv, void* d_identity,
                         GateGradDev dg, float* dbn_w, float* dbn_b, float* dbn2_w, float* dbn2_b, void* workspace,
                         size_t workspace_bytes, hipStream_t stream);

}  // namespace cnsn
