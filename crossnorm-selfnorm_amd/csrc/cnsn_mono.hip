// Channel-in-registers strategy: eligibility, geometry, launches.
#include "cnsn_mono.h"
#include "cnsn_env.h"

#include <cstdlib>

#include "cnsn_mono_kernels.h"

namespace cnsn {

namespace {

// f(TypeTag<T>, IntTag<VEC>, IntTag<LPP>, IntTag<RMAX>)
template <typename F>
bool dispatch_m(int dtype, int vec, int lpp, int rmax, F&& f) {
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
        if (vec == 2) return by_l(TypeTag<_Float16>{}, IntTag<2>{});
        if (vec == 1) return by_l(TypeTag<_Float16>{}, IntTag<1>{});
    }
    return false;
}

// one workgroup per channel, rounded up to whole rounds of the 8 XCDs (MonoWalk)
inline int mono_grid(int C) { return ((C + 7) / 8) * 8; }

MonoArgs make_mono_args(const Plan& pl, const MonoPlan& mp) {
    MonoArgs ma;
    ma.mid = pl.mid;
    ma.nvec = pl.mid.M / mp.vec;
    ma.R = mp.R;
    return ma;
}

}  // namespace

MonoPlan mono_plan(const Plan& pl, int add, bool backward) {
    MonoPlan mp{false, 0, 0, 0, 0, 0};
    const cnsn_problem_t& p = pl.pr;
    if (p.strategy != CNSN_STRATEGY_AUTO && p.strategy != CNSN_STRATEGY_MONO) return mp;
    if (p.cn_active || !p.sn_active || p.sn_two || add == ADD_POST) return mp;  // (two-gate form: never used by a caller)
    if (const char* e = knob(K_MONO))
        if (e[0] == '0' && p.strategy == CNSN_STRATEGY_AUTO) return mp;
    const int b = elem_bytes(p.dtype), M = p.H * p.W;
    // widest vector (16 .. 2 bytes) that divides the plane; a 16-byte vector that would leave fewer than 16 vectors
    // per plane is halved once (more lanes of the slot row busy)
    int vec = 16 / b;
    while (vec > 1 && M % vec) vec >>= 1;
    if (vec * b == 16 && M / vec < 16 && (M / (vec / 2)) <= 64) vec >>= 1;
    const int nvec = M / vec;
    if (nvec > 64 || nvec < 2) return mp;
    const int lpp = nvec <= 16 ? 16 : 64, ppr = 64 / lpp;
    const int rows = (p.N + ppr - 1) / ppr;                   // slot rows the channel needs
    const int R = (rows + kMonoWaves - 1) / kMonoWaves;       // per wave
    if (R > 16 || p.N > kMonoBlock) return mp;
    const int rmax = R <= 8 ? 8 : 16;
    // registers: RMAX slots of vec*b bytes per tensor held (x forward; G and x backward) out of 128 VGPRs per lane
    // (the backward of the 16-row classes walks its rows in parts — halves with 8-byte vectors, quarters with 16-byte ones:
    //  mono_bwd_kernel's NH_ — and holds one part at a time)
    // Quarters: 16-bit tensors only — (256,512,16,16) bf16 backward 0.081 -> 0.053 ms against the two-pass kernels it ran before
    // (16 x 16-byte rows of G and x did not fit at all); in fp32 the cluster kernels of cnsn_resident_sn_kernels.h are ahead
    // of the quarter-wise mono backward (14x14: 0.145 vs 0.156 ms, 16x16: 0.086 vs 0.098), so fp32 stays as it was.
    const bool parts_off = [] { const char* e = knob(K_MONO_RELOAD); return e && e[0] == '0'; }();  // (A/B runs)
    const int parts = (backward && rmax == 16 && !parts_off) ? (vec * b <= 8 ? 2 : (b == 2 ? 4 : 1)) : 1;
    const int data_regs = rmax * vec * b / 4 * (backward ? 2 : 1) / parts;
    if (data_regs > 72) return mp;
    if ((long long)ppr * p.C * M * b >= 0x7ffffff0ll) return mp;   // 31-bit lane offsets inside a slot row
    if (p.strategy == CNSN_STRATEGY_AUTO) {
        // a plane of a few dozen bytes is better served by the channel-local kernels (several channels per workgroup)
        if ((long long)M * b < 64) return mp;
        // one 16-bit element per lane (7x7 bf16: 25 KB per workgroup) does not amortise a 1024-thread workgroup: the
        // channel-local kernels with several channels per workgroup are as fast or faster there (measured: forward 0.074
        // vs 0.075, backward 0.127 vs 0.116 ms at (256,2048,7,7); fp32 0.088 / 0.165 vs 0.112 / 0.177: mono)
        if (vec * b < 4) return mp;
        if (p.N < 16) return mp;  // hardly any planes per channel: the two-pass kernels have more parallelism
    }
    mp.vec = vec;
    mp.lpp = lpp;
    mp.rmax = rmax;
    mp.R = R;
    mp.lds = mono_lds_bytes(kMonoWaves * R * ppr, backward);
    if (mp.lds > 64 * 1024) return mp;  // (N close to 1024 with 16 lanes per plane, backward: not worth a larger LDS grant)
    mp.ok = true;
    return mp;
}

int mono_forward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* x, const void* addend, GateDev g,
                 GateDev f, void* y, double* saved, hipStream_t stream) {
    const MonoArgs ma = make_mono_args(pl, mp);
    const bool epi = add == ADD_PRE || relu;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_m(pl.pr.dtype, mp.vec, mp.lpp, mp.rmax, [&](auto tt, auto vt, auto lt, auto rt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value, RMAX = decltype(rt)::value;
        if (epi) {
            auto kern = mono_fwd_kernel<T, VEC, LPP, RMAX, true>;
            kern<<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(
                ma, (const T*)x, (const T*)(add == ADD_PRE ? addend : nullptr), (T*)y, g, f, saved, add, relu, TailDev{});
        } else {
            auto kern = mono_fwd_kernel<T, VEC, LPP, RMAX, false>;
            kern<<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(ma, (const T*)x, nullptr, (T*)y, g, f, saved,
                                                                                ADD_NONE, 0, TailDev{});
        }
        const hipError_t e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    });
    return status;
}

int mono_backward(const Plan& pl, const MonoPlan& mp, int add, int relu, const void* gy, const void* x, const void* addend,
                  GateDev g, GateDev f, const double* saved, void* dx, GateGradDev dg, GateGradDev df, hipStream_t stream) {
    const MonoArgs ma = make_mono_args(pl, mp);
    const bool epi = add == ADD_PRE || relu;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_m(pl.pr.dtype, mp.vec, mp.lpp, mp.rmax, [&](auto tt, auto vt, auto lt, auto rt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value, RMAX = decltype(rt)::value;
        // sixteen slot rows of G and x do not leave registers for a second workgroup on the CU: two-phase variant (RELOAD)
        // (halves with 8-byte vectors, quarters with 16-byte ones in 16 bits: 32 data registers either way — mono_plan admits the
        //  call on that basis)
        constexpr int kParts = RMAX != 16 ? 1 : (VEC * (int)sizeof(T) <= 8 ? 2 : (sizeof(T) == 2 ? 4 : 1));
        constexpr bool kReloadFits = kParts > 1;
        bool reload = kReloadFits;
        if (const char* e = knob(K_MONO_RELOAD)) reload = reload && e[0] != '0';
        auto launch = [&](auto kern) {
            kern<<<mono_grid(pl.pr.C), kMonoBlock, mp.lds, stream>>>(ma, (const T*)gy, (const T*)x,
                                                                                (const T*)(epi && add == ADD_PRE ? addend : nullptr),
                                                                                (T*)dx, g, f, dg, df, saved, epi ? add : ADD_NONE,
                                                                                epi ? relu : 0, TailDev{});
        };
        if constexpr (kReloadFits) {
            if (reload) {
                if (epi)
                    launch(mono_bwd_kernel<T, VEC, LPP, RMAX, true, false, kParts>);
                else
                    launch(mono_bwd_kernel<T, VEC, LPP, RMAX, false, false, kParts>);
            } else if (epi) {
                launch(mono_bwd_kernel<T, VEC, LPP, RMAX, true>);
            } else {
                launch(mono_bwd_kernel<T, VEC, LPP, RMAX, false>);
            }
        } else if (epi) {
            launch(mono_bwd_kernel<T, VEC, LPP, RMAX, true>);
        } else {
            launch(mono_bwd_kernel<T, VEC, LPP, RMAX, false>);
        }
        const hipError_t e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    });
    return status;
}

}  // namespace cnsn
