// Channel-local strategy: eligibility, geometry, launches.
#include "cnsn_local.h"

#include <cstdlib>

#include "cnsn_local_kernels.h"

namespace cnsn {

namespace {

constexpr size_t kLocalLdsCap = 128 * 1024;  // of the 160 KiB a gfx950 workgroup may have

size_t fwd_lds(int N, int CG, int M, int b) {
    return local_align((size_t)N * CG * M * b) + local_align((size_t)2 * N * CG * 4) + kLocalMaxCG * 16 * 8 + 4 * 4 * 8;
}
size_t bwd_lds(int N, int CG, int M, int b) {
    return 2 * local_align((size_t)N * CG * M * b) + local_align((size_t)4 * N * CG * 4) + (size_t)2 * N * 8 + 4 * 4 * 8;
}

template <typename Kern>
bool allow_lds(Kern kern, size_t lds) {
    if (lds <= 64 * 1024) return true;
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
}

LocalArgs make_local_args(const Plan& pl, const LocalPlan& lp) {
    LocalArgs la;
    la.mid = pl.mid;
    la.CG = lp.CG;
    la.W = lp.W;
    la.piece_vecs = lp.CG * pl.mid.M * elem_bytes(pl.pr.dtype) / lp.W;
    la.planes = pl.pr.N * lp.CG;
    la.inv_m = 1.0f / (float)pl.mid.M;
    return la;
}

// f(TypeTag<T>, IntTag<W>)
template <typename F>
bool dispatch_l(int dtype, int W, F&& f) {
    auto by_w = [&](auto tt) -> bool {
        using T = typename decltype(tt)::type;
        switch (W) {
            case 16: f(tt, IntTag<16>{}); return true;
            case 8: f(tt, IntTag<8>{}); return true;
            case 4: f(tt, IntTag<4>{}); return true;
            case 2:
                if constexpr (sizeof(T) == 2) {
                    f(tt, IntTag<2>{});
                    return true;
                }
                return false;
            default: return false;
        }
    };
    if (dtype == CNSN_F32) return by_w(TypeTag<float>{});
    if (dtype == CNSN_BF16) return by_w(TypeTag<bf16_t>{});
    return by_w(TypeTag<_Float16>{});
}

}  // namespace

LocalPlan local_plan(const Plan& pl, int add, bool backward) {
    LocalPlan lp{false, 0, 0, 0};
    const cnsn_problem_t& p = pl.pr;
    if (p.strategy != CNSN_STRATEGY_AUTO && p.strategy != CNSN_STRATEGY_LOCAL) return lp;
    if (p.cn_active || !p.sn_active || add == ADD_POST) return lp;
    const int b = elem_bytes(p.dtype), M = p.H * p.W;
    const long long plane_bytes = (long long)M * b;
    if (plane_bytes > 1024 || M < 2) return lp;
    // channels per workgroup (CG): more channels = longer contiguous pieces = wider copy vectors, but a larger
    // LDS image = fewer workgroups per CU — and the workgroups of a CU overlapping their load / compute / store
    // phases is what hides the latencies here.  Measured (profiles/r01_small_planes.md): the smallest group that
    // reaches 4-byte vectors wins ((256,2048,7,7) bf16 forward: CG 1/2/4 = 0.141/0.092/0.125 ms), and an image
    // over 64 KiB (one workgroup per CU) loses to the other strategies.  CNSN_LOCAL_CG overrides (tuning).
    int CG = 0, W = 0;
    const char* force = getenv("CNSN_LOCAL_CG");
    for (int cg : {1, 2, 4, 8}) {
        if (p.C % cg || (force && atoi(force) != cg)) continue;
        const size_t need = backward ? bwd_lds(p.N, cg, M, b) : fwd_lds(p.N, cg, M, b);
        if (need > kLocalLdsCap || (long long)p.N * cg > 4096) continue;
        const long long piece = (long long)cg * plane_bytes, row = (long long)p.C * plane_bytes;
        int w = 16;
        while (w > b && (piece % w || row % w)) w >>= 1;
        if (piece % w || row % w) continue;
        if (CG == 0 || (w > W && W < 4)) {
            CG = cg;
            W = w;
            lp.lds = need;
        }
    }
    if (CG == 0) return lp;
    if (p.strategy == CNSN_STRATEGY_AUTO && (lp.lds > 64 * 1024 || W < 4)) return lp;
    lp.CG = CG;
    lp.W = W;
    lp.ok = true;
    return lp;
}

int local_forward(const Plan& pl, const LocalPlan& lp, int add, int relu, const void* x, const void* addend, GateDev g,
                  GateDev f, void* y, double* saved, hipStream_t stream) {
    const LocalArgs la = make_local_args(pl, lp);
    const bool epi = add == ADD_PRE || relu;
    const int grid = pl.pr.C / lp.CG;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_l(pl.pr.dtype, lp.W, [&](auto tt, auto wt) {
        using T = typename decltype(tt)::type;
        constexpr int W = decltype(wt)::value;
        auto go = [&](auto kern) {
            if (!allow_lds(kern, lp.lds)) return;
            kern<<<grid, kBlock, lp.lds, stream>>>(la, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), (T*)y, g, f,
                                                  saved, relu);
            const hipError_t e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        if (epi)
            go(local_fwd_kernel<T, W, true>);
        else
            go(local_fwd_kernel<T, W, false>);
    });
    return status;
}

int local_backward(const Plan& pl, const LocalPlan& lp, int add, int relu, const void* gy, const void* x,
                   const void* addend, GateDev g, GateDev f, const double* saved, void* dx, GateGradDev dg, GateGradDev df,
                   hipStream_t stream) {
    const LocalArgs la = make_local_args(pl, lp);
    const bool epi = add == ADD_PRE || relu;
    const int grid = pl.pr.C / lp.CG;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_l(pl.pr.dtype, lp.W, [&](auto tt, auto wt) {
        using T = typename decltype(tt)::type;
        constexpr int W = decltype(wt)::value;
        auto go = [&](auto kern) {
            if (!allow_lds(kern, lp.lds)) return;
            kern<<<grid, kBlock, lp.lds, stream>>>(la, (const T*)gy, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr),
                                                  (T*)dx, g, f, dg, df, saved, relu);
            const hipError_t e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        if (epi)
            go(local_bwd_kernel<T, W, true>);
        else
            go(local_bwd_kernel<T, W, false>);
    });
    return status;
}

}  // namespace cnsn
