// SelfNorm-only cluster kernels, backward: host entry point (shared logic in cnsn_resident_sn_host.h).
#include "cnsn_resident_sn_host.h"
#include "cnsn_env.h"

namespace cnsn {

int resident_sn_backward(const cnsn_problem_t& p, const MidArgs& mid, int add, int relu, const void* gy, const void* x,
                         const void* addend, GateDev g, const double* saved, void* dx, GateGradDev dg, void* workspace,
                         hipStream_t stream) {
    const SnxPlan sp = snxhost::plan_impl(p, false, add, relu, true);
    if (!sp.ok) return CNSN_E_UNSUPPORTED;
    const bool epi = add != ADD_NONE || relu;
    ResArgs ra = snxhost::make_args(p, mid, sp);
#ifdef CNSN_PROF  // tuning builds: time stamps land 4 MiB into the workspace (callers size it accordingly)
    if (knob(K_PROF)) ra.prof = (unsigned long long*)((char*)workspace + (4u << 20));
#endif
    const size_t lds = snxhost::lds_bytes(sp.K, 4 * sp.ppw, sp.npark, true, sp.vec * elem_bytes(p.dtype));
    int status = CNSN_E_UNSUPPORTED;
    auto run = [&](auto tt, auto vt, auto nt, auto pt, auto et) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value, PPW = decltype(pt)::value;
        constexpr bool EPI = decltype(et)::value != 0;
        auto kern = resident_sn_bwd_kernel<T, VEC, NV, PPW, EPI>;
        if (!allow_dynamic_lds(kern, lds)) return;
        const int grid = reshost::grid_for(kern, lds, sp.K, ra.items);
        if (grid < sp.K) return;
        ResidentChain chain(stream);
        const ExchangeArea ea = resident_exchange_area(p, snxhost::tagged_bytes(p, sp.K, true), workspace, stream, true);
        ra.epoch = ea.epoch;
        ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
        unsigned* ctl = (unsigned*)ea.base;
        unsigned long long* gran = (unsigned long long*)((char*)ea.base + kCtlBytes);
        // round B behind round A (+ 256: the scalar-path gather of round A reads whole 256-byte groups)
        const size_t a_gran = (size_t)p.C * sp.K * (ea.epoch ? 4 : 2);
        unsigned long long* gran_b = gran + ((a_gran + 32 + 31) & ~(size_t)31);
        const size_t b_gran = (size_t)p.C * sp.K * 4 * (ea.epoch ? 2 : 1);
        hipError_t e = ea.epoch ? hipSuccess
                                : hipMemsetAsync(workspace, 0xff, (size_t)((char*)(gran_b + b_gran) - (char*)workspace), stream);
        if (e != hipSuccess) {
            status = (int)e;
            return;
        }
        const SnxBwdKargs<T> ka{ra, sp.npark, (const T*)gy, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), relu,
                                (T*)dx, g, dg, gran, gran_b, saved, ctl};
        kern<<<grid, kBlock, lds, stream>>>(ka);
        e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    };
    if (epi)
        snxhost::dispatch_snx<true, true>(p.dtype, sp.vec, sp.nv,
                                          [&](auto tt, auto vt, auto nt, auto pt) { run(tt, vt, nt, pt, IntTag<1>{}); });
    else
        snxhost::dispatch_snx<true, false>(p.dtype, sp.vec, sp.nv,
                                           [&](auto tt, auto vt, auto nt, auto pt) { run(tt, vt, nt, pt, IntTag<0>{}); });
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] sn-cluster bwd: nv=%d ppw=%d K=%d npark=%d epi=%d lds=%zu -> status %d\n", sp.nv, sp.ppw, sp.K,
                sp.npark, (int)epi, lds, status);
    return status;
}

SnxPlan resident_sn_cn_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm) {
    if (has_chan_perm) return SnxPlan{false, 0, 0, 0, 0, 0};
    return snxhost::plan_impl(p, boxed, ADD_NONE, 0, true, true);
}

int resident_sn_cn_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy, const void* x,
                            const int64_t* perm, GateDev g, const double* saved, void* dx, GateGradDev dg, void* workspace,
                            hipStream_t stream) {
    const SnxPlan sp = snxhost::plan_impl(p, boxed, ADD_NONE, 0, true, true);
    if (!sp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    ResArgs ra = reshost::make_args(p, cb, sb, mid, ResPlan{true, sp.vec, sp.nv, sp.ppw, sp.K});
#ifdef CNSN_PROF
    if (knob(K_PROF)) ra.prof = (unsigned long long*)((char*)workspace + (4u << 20));
#endif
    const size_t lds = snxhost::lds_bytes(sp.K, 4 * sp.ppw, sp.npark, true, sp.vec * elem_bytes(p.dtype), true, p.N);
    int status = CNSN_E_UNSUPPORTED;
    snxhost::dispatch_snx<true, false>(p.dtype, sp.vec, sp.nv, [&](auto tt, auto vt, auto nt, auto pt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value, PPW = decltype(pt)::value;
        if constexpr (snxhost::snx_cn_built(NV, (int)sizeof(T), VEC * (int)sizeof(T))) {
            auto launch = [&](auto kern) {
                if (!allow_dynamic_lds(kern, lds)) return;
                const int grid = reshost::grid_for(kern, lds, sp.K, ra.items);
                if (grid < sp.K) return;
                ResidentChain chain(stream);
                const ExchangeArea ea =
                    resident_exchange_area(p, snxhost::tagged_bytes(p, sp.K, true, true, boxed), workspace, stream, true);
                ra.epoch = ea.epoch;
                ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
                unsigned* ctl = (unsigned*)ea.base;
                unsigned long long* gran = (unsigned long long*)((char*)ea.base + kCtlBytes);
                const size_t a_gran = (size_t)p.C * sp.K * (ea.epoch ? 4 : 2);
                unsigned long long* gran_b = gran + ((a_gran + 32 + 31) & ~(size_t)31);
                const size_t b_gran = (size_t)p.C * sp.K * 4 * (ea.epoch ? 2 : 1);
                unsigned long long* gran_p = gran_b + ((b_gran + 31) & ~(size_t)31);  // per-plane sums behind round B
                const size_t p_gran = (size_t)p.N * p.C * (ea.epoch ? 2 : 1) * (boxed ? 2 : 1);
                hipError_t e = ea.epoch ? hipSuccess
                                        : hipMemsetAsync(workspace, 0xff, (size_t)((char*)(gran_p + p_gran) - (char*)workspace), stream);
                if (e != hipSuccess) {
                    status = (int)e;
                    return;
                }
                SnxBwdKargsCn<T> ka{{ra, sp.npark, (const T*)gy, (const T*)x, nullptr, 0, (T*)dx, g, dg, gran, gran_b, saved, ctl},
                                    gran_p, perm, *pin};
                kern<<<grid, kBlock, lds, stream>>>(ka);
                e = hipGetLastError();
                status = e == hipSuccess ? CNSN_OK : (int)e;
            };
            if (boxed)
                launch(resident_sn_bwd_kernel<T, VEC, NV, PPW, false, true, true>);
            else
                launch(resident_sn_bwd_kernel<T, VEC, NV, PPW, false, true, false>);
        }
    });
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] sn-cluster bwd (CrossNorm): nv=%d ppw=%d K=%d npark=%d lds=%zu -> status %d\n", sp.nv, sp.ppw, sp.K,
                sp.npark, lds, status);
    return status;
}

}  // namespace cnsn
