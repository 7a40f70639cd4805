// Cluster-resident kernels for SelfNorm alone (cnsn_resident_sn_kernels.h): host entry points.
#pragma once
#include "cnsn_host_plan.h"
#include "cnsn_resident_kernels.h"

namespace cnsn {

struct SnxPlan {
    bool ok;
    int vec, nv, ppw, K, npark;
};
// add: ADD_NONE or ADD_PRE; relu 0/1.  ok only for SelfNorm alone in training mode (one gate, no CrossNorm).
SnxPlan resident_sn_plan(const cnsn_problem_t& p, bool boxed, int add, int relu, bool backward);
// AUTO only: these kernels go BEFORE the channel-in-registers / local strategies the entry points try first (measured
// classes of one-slot planes: profiles/r03_sn_cluster.md)
bool resident_sn_prefers(const cnsn_problem_t& p, bool boxed, int add, int relu, bool backward);
// bytes of exchange area a launch may need (the persistent context is sized for it, cnsn_context_bytes)
size_t resident_sn_exchange_bytes(const cnsn_problem_t& p);

int resident_sn_forward(const cnsn_problem_t& p, const MidArgs& mid, int add, int relu, const void* x, const void* addend,
                        GateDev g, void* y, double* saved, void* workspace, hipStream_t stream);
int resident_sn_backward(const cnsn_problem_t& p, const MidArgs& mid, int add, int relu, const void* gy, const void* x,
                         const void* addend, GateDev g, const double* saved, void* dx, GateGradDev dg, void* workspace,
                         hipStream_t stream);

// Round 4 — the backward of un-boxed CrossNorm + SelfNorm (models/cnsn.py:58-91 with crop='neither' in front of :130-150; no
// channel permutation, one gate, no epilogue) in the partial-moment cluster kernels: ok only for that call.  `perm`: device
// array or NULL with cnsn_problem_t.perm_host (launch argument).
SnxPlan resident_sn_cn_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm);
int resident_sn_cn_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy, const void* x,
                            const int64_t* perm, GateDev g, const double* saved, void* dx, GateGradDev dg, void* workspace,
                            hipStream_t stream);

}  // namespace cnsn
