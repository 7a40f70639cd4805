// Packed small-plane kernels: host entry points (kernels in cnsn_packed_kernels.h, built by cnsn_packed.hip).
// `add` is an AddMode of cnsn_fused_stream_kernels.h (0 none, 1 pre, 2 post); the op alone passes 0 / relu 0.
#pragma once
#include "cnsn_host_plan.h"

namespace cnsn {

struct PackedGeom {
    int P, M, Wd;     // planes, elements per plane, width
    int R;            // planes per run
    int runs;         // ceil(P / R)
    int run_vecs;     // 16-byte vectors per run
    long long total;  // bytes in the tensor
    Box cb, sb;
};

// can (and should) this problem's tensor passes run on the packed kernels?  fills `g`
bool packed_plan(const Plan& pl, PackedGeom& g);

void packed_stats(const Plan& pl, const PackedGeom& g, int add, const void* x, const void* addend, double* mom,
                  hipStream_t stream);
void packed_apply_fwd(const Plan& pl, const PackedGeom& g, int add, int relu, const void* x, const void* addend, void* y,
                      const float* coef, hipStream_t stream);
void packed_reduce(const Plan& pl, const PackedGeom& g, int add, int relu, const void* gy, const void* x,
                   const void* addend, const double* saved, float* sums, hipStream_t stream);
void packed_apply_bwd(const Plan& pl, const PackedGeom& g, int add, int relu, const void* gy, const void* x,
                      const void* addend, void* dx, void* d_addend, const float* coef, const double* saved,
                      hipStream_t stream);

}  // namespace cnsn
