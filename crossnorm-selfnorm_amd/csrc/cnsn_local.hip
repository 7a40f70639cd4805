// Channel-local strategy: eligibility, geometry, launches.
#include "cnsn_local.h"
#include "cnsn_env.h"

#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "cnsn_local_kernels.h"

namespace cnsn {

namespace {

constexpr size_t kLocalLdsCap = 128 * 1024;  // of the 160 KiB a gfx950 workgroup may have

// threads per workgroup: 256 for a small image (several workgroups per CU overlap their phases), 1024 for a large one
// (measured, profiles/r01_small_planes.md: 256 threads win whenever two or more workgroups fit a CU — images up to
// 64 KiB —, 1024 threads when a workgroup is alone on its CU)
int block_for(size_t lds256) {
    if (const char* e = knob(K_LOCAL_LB)) return atoi(e) == 256 ? 256 : kLocalBigBlock;
    return lds256 > 64 * 1024 ? kLocalBigBlock : 256;
}
size_t fwd_lds_lb(int N, int CG, int M, int b, int LB) {
    return local_align((size_t)N * CG * M * b) + local_align((size_t)2 * N * CG * 4) + kLocalMaxCG * 16 * 8 + (size_t)(LB / 64) * 4 * 8;
}
size_t bwd_lds_lb(int N, int CG, int M, int b, int LB) {
    return 2 * local_align((size_t)N * CG * M * b) + local_align((size_t)4 * N * CG * 4) + (size_t)(LB / 256) * 2 * N * 8 +
           (size_t)(LB / 64) * 4 * 8;
}
size_t fwd_lds(int N, int CG, int M, int b) { return fwd_lds_lb(N, CG, M, b, block_for(fwd_lds_lb(N, CG, M, b, 256))); }
size_t bwd_lds(int N, int CG, int M, int b) { return bwd_lds_lb(N, CG, M, b, block_for(bwd_lds_lb(N, CG, M, b, 256))); }

// more than 64 KiB of dynamic LDS has to be allowed per kernel and device once (idempotent; remembered so that the
// launch path stays free of runtime calls)
template <typename Kern>
bool allow_lds(Kern kern, size_t lds) {
    if (lds <= 64 * 1024) return true;
    static std::mutex mu;
    static std::unordered_map<uintptr_t, size_t> allowed;  // per (device, kernel): the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[(uintptr_t)(const void*)kern * 31u + (uintptr_t)dev];
    if (have >= lds) return true;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLocalLdsCap) != hipSuccess)
        return false;
    have = kLocalLdsCap;
    return true;
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
    LocalPlan lp{false, 0, 0, 0, 256};
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
    const char* force = knob(K_LOCAL_CG);
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
    // AUTO: copies of at least 4 bytes; an image over 64 KiB (a workgroup alone on its CU) only where the two-pass
    // alternative is slower (not fp32 with the channel-tiled mid kernels of C >= 512) and only where the cluster-resident kernels cannot take the plane (fewer than 33 vectors, or no 8/16-byte
    // vectors) — (256,2048,7,7) backward: bf16 0.117 vs 0.127 ms two-pass, but fp32 0.203 vs 0.163 once the mid
    // kernels are channel-tiled; where the resident kernels can take the plane they are as fast or faster
    if (p.strategy == CNSN_STRATEGY_AUTO) {
        const int rv = pick_vec(p.dtype, M);
        const bool resident_can = (rv * b == 16 || (b == 2 && rv == 4)) && M / rv > 32;
        if (W < 4 || (lp.lds > 64 * 1024 && (resident_can || (b == 4 && p.C >= 512)))) return lp;
    }
    lp.CG = CG;
    lp.W = W;
    lp.block = block_for(backward ? bwd_lds_lb(p.N, CG, M, b, 256) : fwd_lds_lb(p.N, CG, M, b, 256));
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
        auto go = [&](auto kern, int lb) {
            if (!allow_lds(kern, lp.lds)) return;
            kern<<<grid, lb, lp.lds, stream>>>(la, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), (T*)y, g, f, saved,
                                             relu);
            const hipError_t e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        if (lp.block == 256) {
            if (epi)
                go(local_fwd_kernel<T, W, true, 256>, 256);
            else
                go(local_fwd_kernel<T, W, false, 256>, 256);
        } else {
            if (epi)
                go(local_fwd_kernel<T, W, true, kLocalBigBlock>, kLocalBigBlock);
            else
                go(local_fwd_kernel<T, W, false, kLocalBigBlock>, kLocalBigBlock);
        }
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
        auto go = [&](auto kern, int lb) {
            if (!allow_lds(kern, lp.lds)) return;
            kern<<<grid, lb, lp.lds, stream>>>(la, (const T*)gy, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr),
                                             (T*)dx, g, f, dg, df, saved, relu);
            const hipError_t e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        if (lp.block == 256) {
            if (epi)
                go(local_bwd_kernel<T, W, true, 256>, 256);
            else
                go(local_bwd_kernel<T, W, false, 256>, 256);
        } else {
            if (epi)
                go(local_bwd_kernel<T, W, true, kLocalBigBlock>, kLocalBigBlock);
            else
                go(local_bwd_kernel<T, W, false, kLocalBigBlock>, kLocalBigBlock);
        }
    });
    return status;
}

}  // namespace cnsn
