// Channel-in-registers strategy with CrossNorm: eligibility, geometry, launches (kernels: cnsn_mono_cn_kernels.h).
#include <cstdlib>
#include "cnsn_env.h"

#include "cnsn_mono.h"
#include "cnsn_mono_cn_kernels.h"

namespace cnsn {

namespace {

// f(TypeTag<T>, IntTag<VEC>, IntTag<LPP>, IntTag<RMAX>)
template <typename F>
bool dispatch_mc(int dtype, int vec, int lpp, int rmax, F&& f) {
    auto by_r = [&](auto tt, auto vt, auto lt) -> bool {
        if (rmax == 8) {
            f(tt, vt, lt, IntTag<8>{});
            return true;
        }
        if (rmax == 16) {
            f(tt, vt, lt, IntTag<16>{});
            return true;
        }
        return false;
    };
    auto by_l = [&](auto tt, auto vt) -> bool {
        if (lpp == 16) return by_r(tt, vt, IntTag<16>{});
        if (lpp == 64) return by_r(tt, vt, IntTag<64>{});
        return false;
    };
    if (dtype == CNSN_F32) {
        if (vec == 4) return by_l(TypeTag<float>{}, IntTag<4>{});
        if (vec == 2) return by_l(TypeTag<float>{}, IntTag<2>{});
        if (vec == 1) return by_l(TypeTag<float>{}, IntTag<1>{});
    } else if (dtype == CNSN_BF16) {
        if (vec == 8) return by_l(TypeTag<bf16_t>{}, IntTag<8>{});
        if (vec == 4) return by_l(TypeTag<bf16_t>{}, IntTag<4>{});
        if (vec == 2) return by_l(TypeTag<bf16_t>{}, IntTag<2>{});
        if (vec == 1) return by_l(TypeTag<bf16_t>{}, IntTag<1>{});
    } else if (dtype == CNSN_F16) {
        if (vec == 8) return by_l(TypeTag<_Float16>{}, IntTag<8>{});
        if (vec == 4) return by_l(TypeTag<_Float16>{}, IntTag<4>{});
    }
    return false;
}

MonoCnArgs make_cn_args(const Plan& pl, const MonoPlan& mp) {
    MonoCnArgs ca;
    ca.m.mid = pl.mid;
    ca.m.nvec = pl.mid.M / mp.vec;
    ca.m.R = mp.R;
    ca.Wd = pl.pr.W;
    ca.cb = pl.cb;
    ca.sb = pl.sb;
    return ca;
}

inline int mono_grid(int C) { return ((C + 7) / 8) * 8; }

}  // namespace

MonoPlan mono_cn_plan(const Plan& pl, bool has_chan_perm, int add, bool backward) {
    MonoPlan mp{false, 0, 0, 0, 0, 0};
    const cnsn_problem_t& p = pl.pr;
    if (p.strategy != CNSN_STRATEGY_AUTO && p.strategy != CNSN_STRATEGY_MONO) return mp;
    if (!p.cn_active || has_chan_perm || (p.sn_active && p.sn_two) || add == ADD_POST) return mp;
    if (const char* e = knob(K_MONO))
        if (e[0] == '0' && p.strategy == CNSN_STRATEGY_AUTO) return mp;
    const int b = elem_bytes(p.dtype), M = p.H * p.W;
    int vec = 16 / b;  // masks are per element: a vector may straddle rows of the plane
    while (vec > 1 && M % vec) vec >>= 1;
    if (vec * b == 16 && M / vec < 16 && (M / (vec / 2)) <= 64) vec >>= 1;
    if (p.dtype == CNSN_F16 && vec < 4) return mp;  // (not instantiated)
    const int nvec = M / vec;
    if (nvec > 64 || nvec < 2 || M < 2) return mp;
    const int lpp = nvec <= 16 ? 16 : 64, ppr = 64 / lpp;
    const int rows = (p.N + ppr - 1) / ppr;
    const int R = (rows + kMonoWaves - 1) / kMonoWaves;
    if (R > 16 || p.N > kMonoBlock) return mp;
    const int rmax = R <= 8 ? 8 : 16;
    const int data_regs = rmax * (vec * b < 4 ? 4 : vec * b) / 4 * (backward ? 2 : 1);
    if (data_regs > 72) return mp;
    if ((long long)ppr * p.C * M * b >= 0x7ffffff0ll) return mp;
    mp.lds = mono_cn_lds_bytes(kMonoWaves * R * ppr, backward);
    if (mp.lds > 64 * 1024) return mp;
    if (p.strategy == CNSN_STRATEGY_AUTO) {
        if ((long long)M * b < 64 || p.N < 16) return mp;
        // one element per lane (7x7) only pays with crop boxes — measured at (256,2048,7,7), fwd+bwd: bf16 boxed 0.305 vs
        // 0.361 ms packed two-pass, un-boxed 0.286 vs 0.258; fp32 un-boxed 0.299 vs 0.289, at N = 96 0.162 vs 0.132
        if (vec * b <= 4 && !pl.boxed) return mp;
        // un-boxed calls the cluster kernels can take: with fewer than ~96 planes per channel one workgroup per channel is
        // under-filled and the cluster kernels win — (64,1024,14,14) bf16 0.099 vs 0.067 ms, fp32 0.107 vs 0.088 on two
        // boxes; at N = 96 / 128 the two are within noise of each other, at 256 mono leads (profiles/r02_auto_audit.md)
        // (round 4, tools/auto_audit.py on two boxes: at N = 96 the cluster kernels lead by 10 % in fp32 and 19 % in bf16 at
        //  (96,1024,14,14); at N = 128 by 5 % at (128,64,16,16) fp32 — the threshold moved from 96 to 128)
        if (!pl.boxed && p.N < 128 && resident_plan(p, false, has_chan_perm, backward).ok) return mp;
        // (round 5, three audits on three boxes: with 64 channels — 64 workgroups for 256 compute units — the cluster kernels lead
        //  at N = 128 too, with and without crop boxes, in fp32: (128,64,16,16) 0.065-0.068 vs 0.072-0.073 ms un-boxed, 0.078-0.079
        //  vs 0.085-0.088 boxed; in bf16 the two are level)
        if (p.dtype == CNSN_F32 && p.C <= 64 && p.N <= 128 && resident_plan(p, pl.boxed, has_chan_perm, backward).ok) return mp;
    }
    mp.vec = vec;
    mp.lpp = lpp;
    mp.rmax = rmax;
    mp.R = R;
    mp.ok = true;
    return mp;
}

int mono_cn_forward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* x, const void* addend,
                    const int64_t* perm, GateDev g, void* y, double* saved, hipStream_t stream) {
    const MonoCnArgs ca = make_cn_args(pl, mp);
    const bool epi = add == ADD_PRE || relu;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_mc(pl.pr.dtype, mp.vec, mp.lpp, mp.rmax, [&](auto tt, auto vt, auto lt, auto rt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value, RMAX = decltype(rt)::value;
        if (epi)
            mono_cn_fwd_kernel<T, VEC, LPP, RMAX, true><<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(
                ca, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), (T*)y, perm, g, saved, add, relu);
        else
            mono_cn_fwd_kernel<T, VEC, LPP, RMAX, false><<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(
                ca, (const T*)x, nullptr, (T*)y, perm, g, saved, ADD_NONE, 0);
        const hipError_t e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    });
    return status;
}

int mono_cn_backward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* gy, const void* x, const void* addend,
                     const int64_t* perm, GateDev g, const double* saved, void* dx, GateGradDev dg, hipStream_t stream) {
    const MonoCnArgs ca = make_cn_args(pl, mp);
    const bool epi = add == ADD_PRE || relu;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_mc(pl.pr.dtype, mp.vec, mp.lpp, mp.rmax, [&](auto tt, auto vt, auto lt, auto rt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value, RMAX = decltype(rt)::value;
        if (epi)
            mono_cn_bwd_kernel<T, VEC, LPP, RMAX, true><<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(
                ca, (const T*)gy, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), (T*)dx, perm, g, dg, saved, add, relu);
        else
            mono_cn_bwd_kernel<T, VEC, LPP, RMAX, false><<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(
                ca, (const T*)gy, (const T*)x, nullptr, (T*)dx, perm, g, dg, saved, ADD_NONE, 0);
        const hipError_t e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    });
    return status;
}

}  // namespace cnsn
