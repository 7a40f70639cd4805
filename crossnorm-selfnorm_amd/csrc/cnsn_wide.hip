// Channel-group-in-registers strategy: eligibility, geometry, launches.
#include "cnsn_wide.h"
#include "cnsn_env.h"

#include <cstdlib>

#include "cnsn_wide_kernels.h"

namespace cnsn {

namespace {

// f(TypeTag<T>, IntTag<VEC>): VEC elements = VB bytes per lane (16 forward, 8 backward)
template <int VB, typename F>
bool dispatch_w(int dtype, F&& f) {
    if (dtype == CNSN_F32) { f(TypeTag<float>{}, IntTag<VB / 4>{}); return true; }
    if (dtype == CNSN_BF16) { f(TypeTag<bf16_t>{}, IntTag<VB / 2>{}); return true; }
    if (dtype == CNSN_F16) { f(TypeTag<_Float16>{}, IntTag<VB / 2>{}); return true; }
    return false;
}

inline int wide_grid(int groups) { return ((groups + 7) / 8) * 8; }

}  // namespace

WidePlan wide_plan(const Plan& pl, int add, bool backward, bool has_chan_perm) {
    WidePlan wp{false, 0, 0, 0};
    const cnsn_problem_t& p = pl.pr;
    if (p.strategy != CNSN_STRATEGY_AUTO && p.strategy != CNSN_STRATEGY_MONO) return wp;
    if ((!p.sn_active && !p.cn_active) || (p.sn_active && p.sn_two) || add == ADD_POST) return wp;
    // CrossNorm: without crop boxes and without the channel permutation (the pairing stays inside one channel); CrossNorm
    // ALONE (models/cnsn.py:152-164 with selfnorm=None) runs the same kernels without a gate (round 4: it was packed two-pass,
    // 0.203 ms at (256,2048,7,7) bf16 against 0.138 for CrossNorm+SelfNorm here)
    if (p.cn_active && (pl.boxed || has_chan_perm)) return wp;
    if (!p.sn_active && (add != ADD_NONE)) return wp;
    int mode = 1;  // CNSN_WIDE=0: never; CNSN_WIDE=2: wherever eligible (tests: fp32 and small batches too)
    if (const char* e = knob(K_WIDE)) mode = e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1);
    if (mode == 0) return wp;
    const int b = elem_bytes(p.dtype), M = p.H * p.W;
    if (M > 64 || (M * b) % 8 == 0) return wp;  // whole 8-byte vectors: the mono / cluster kernels' ground
    const int vec = 16 / b;  // = channels per workgroup, both directions
    if (M < vec) return wp;                        // a lane may straddle at most two channels
    const int vec_all = 16 / b;
    if (p.C % vec_all) return wp;                  // (the same channel count serves both directions)
    if (p.N > 16 * kWideRows || p.N < 16) return wp;
    if (mode != 2) {
        // measured on MI355X at (N,2048,7,7) (profiles/r02_wide_7x7.md): N = 256 bf16 0.108 vs 0.185 ms per step (channel-
        // local kernels), fp32 0.189 vs 0.224 (mono, 4-byte accesses); N <= 96: the step is host-bound either way
        if (p.N < 128) return wp;
    }
    wp.vec = vec;
    wp.R = (p.N + kWideWaves - 1) / kWideWaves;
    const bool cn = p.cn_active != 0;
    wp.lds = backward ? wide_lds_bytes(p.N, vec, 4, 2, cn ? 10 : 7, 0, cn ? p.N : 0)
                      : wide_lds_bytes(p.N, vec, 8, 1, cn ? 3 : 2, kWideParkFwd * 16, cn ? p.N : 0);
    if (wp.lds > 160 * 1024) return wp;
    wp.ok = true;
    return wp;
}

int wide_forward(const Plan& pl, const WidePlan& wp, int add, int relu, const void* x, const void* addend, const int64_t* perm,
                 GateDev g, void* y, double* saved, hipStream_t stream) {
    WideArgs wa{pl.mid, wp.R};
    const bool epi = add == ADD_PRE || relu;
    const int grid = wide_grid(pl.pr.C / wp.vec);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_w<16>(pl.pr.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        auto launch = [&](auto kern) {
            if (!allow_dynamic_lds(kern, wp.lds)) return;
            kern<<<grid, kWideBlock, wp.lds, stream>>>(wa, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), (T*)y, g,
                                                       saved, epi ? add : ADD_NONE, epi ? relu : 0, perm);
            const hipError_t e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        const bool cn = pl.pr.cn_active != 0;
        if (cn && !perm) {
            status = CNSN_E_NULL;
            return;
        }
        if (epi)
            cn ? launch(wide_fwd_kernel<T, VEC, true, true>) : launch(wide_fwd_kernel<T, VEC, true, false>);
        else
            cn ? launch(wide_fwd_kernel<T, VEC, false, true>) : launch(wide_fwd_kernel<T, VEC, false, false>);
    });
    return status;
}

int wide_backward(const Plan& pl, const WidePlan& wp, int add, int relu, const void* gy, const void* x, const void* addend,
                  const int64_t* perm, GateDev g, const double* saved, void* dx, GateGradDev dg, hipStream_t stream) {
    WideArgs wa{pl.mid, wp.R};
    const bool epi = add == ADD_PRE || relu;
    const int grid = wide_grid(pl.pr.C / wp.vec);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_w<16>(pl.pr.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        auto launch = [&](auto kern) {
            if (!allow_dynamic_lds(kern, wp.lds)) return;
            kern<<<grid, kWideBlock, wp.lds, stream>>>(wa, (const T*)gy, (const T*)x,
                                                       (const T*)(add == ADD_PRE ? addend : nullptr), (T*)dx, g, dg, saved,
                                                       epi ? add : ADD_NONE, epi ? relu : 0, perm);
            const hipError_t e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        const bool cn = pl.pr.cn_active != 0;
        if (cn && !perm) {
            status = CNSN_E_NULL;
            return;
        }
        if (epi)
            cn ? launch(wide_bwd_kernel<T, VEC, true, true>) : launch(wide_bwd_kernel<T, VEC, true, false>);
        else
            cn ? launch(wide_bwd_kernel<T, VEC, false, true>) : launch(wide_bwd_kernel<T, VEC, false, false>);
    });
    return status;
}

}  // namespace cnsn
