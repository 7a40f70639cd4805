// cnsn_forward_bnrelu / cnsn_backward_bnrelu (include/cnsn_hip.h): the op followed by the NEXT block's BatchNorm2d + ReLU
// (models/cifar/wideresnet_cnsn.py:93-96 + :76-77) in one launch per direction — the TAIL instantiations of the
// channel-in-registers kernels (cnsn_mono_kernels.h).
#include "../../include/cnsn_hip.h"

#include <hip/hip_runtime.h>

#include "cnsn_fused_stream_kernels.h"
#include "cnsn_host_plan.h"
#include "cnsn_mono.h"
#include "cnsn_mono_kernels.h"

using namespace cnsn;

namespace {

// f(TypeTag<T>, IntTag<VEC>, IntTag<LPP>, IntTag<RMAX>): the vector widths a whole channel in registers is offered for
template <typename F>
bool dispatch_tail(int dtype, int vec, int lpp, int rmax, F&& f) {
    auto by_r = [&](auto tt, auto vt, auto lt) -> bool {
        if (rmax == 8) {  // (16 slot rows: the backward's third tensor does not fit next to G and X — not built)
            f(tt, vt, lt, IntTag<8>{});
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
    } else if (dtype == CNSN_BF16) {
        if (vec == 8) return by_l(TypeTag<bf16_t>{}, IntTag<8>{});
        if (vec == 4) return by_l(TypeTag<bf16_t>{}, IntTag<4>{});
    } else if (dtype == CNSN_F16) {
        if (vec == 8) return by_l(TypeTag<_Float16>{}, IntTag<8>{});
        if (vec == 4) return by_l(TypeTag<_Float16>{}, IntTag<4>{});
    }
    return false;
}

inline int mono_grid(int C) { return ((C + 7) / 8) * 8; }

struct TailPlan {
    bool ok;
    MonoPlan mp;
    int add;
    const void* addend;
};

int parse(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, Plan& pl, TailPlan& tp, bool backward, bool need_addend) {
    tp = TailPlan{false, MonoPlan{false, 0, 0, 0, 0, 0}, ADD_NONE, nullptr};
    int st = make_plan(prob, pl);
    if (st) return st;
    if (pl.pr.layout != CNSN_LAYOUT_NCHW) return CNSN_OK;  // (not offered for channels-last tensors: tp.ok stays false)
    if (epi) {
        if (epi->struct_bytes != (int32_t)sizeof(cnsn_epilogue_t)) return CNSN_E_STRUCT;
        if (epi->relu || epi->sum_out || !(epi->add_mode == CNSN_ADD_NONE || epi->add_mode == CNSN_ADD_PRE)) return CNSN_OK;  // (not offered)
        tp.add = epi->add_mode;
        tp.addend = epi->addend;
        if (tp.add == ADD_PRE && need_addend) {
            if (!tp.addend) return CNSN_E_NULL;
            if (((uintptr_t)tp.addend & 15u) != 0) return CNSN_E_ALIGN;
        }
    }
    const cnsn_problem_t& p = pl.pr;
    if (p.cn_active || !p.sn_active || p.sn_two) return CNSN_OK;
    MonoPlan mp = mono_plan(pl, tp.add, backward);
    if (!mp.ok) return CNSN_OK;
    if (mp.vec * elem_bytes(p.dtype) < 8) return CNSN_OK;  // (the narrow slots of 7x7 planes: not built for the tail)
    if (mp.rmax != 8) return CNSN_OK;                       // (more than 8 slot rows per wave: not built, see dispatch_tail)
    mp.lds = mono_lds_bytes(kMonoWaves * mp.R * (64 / mp.lpp), backward, true);
    if (mp.lds > 64 * 1024) return CNSN_OK;
    tp.mp = mp;
    tp.ok = true;
    return CNSN_OK;
}

bool tail_ok(const cnsn_bn_tail_t* t) {
    return t && t->struct_bytes == (int32_t)sizeof(cnsn_bn_tail_t) && t->weight && t->bias && t->running_mean && t->running_var;
}

MonoArgs make_args(const Plan& pl, const MonoPlan& mp) {
    MonoArgs ma;
    ma.mid = pl.mid;
    ma.nvec = pl.mid.M / mp.vec;
    ma.R = mp.R;
    return ma;
}

}  // namespace

extern "C" {

int cnsn_bnrelu_plan(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, int backward) {
    Plan pl;
    TailPlan tp;
    const int st = parse(prob, epi, pl, tp, backward != 0, false);
    if (st) return st;
    return tp.ok ? 1 : 0;
}

int cnsn_forward_bnrelu(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* tail, const void* x,
                        const cnsn_gate_t* g, void* y, void* z, float* saved, float* bn_stats, void* workspace,
                        size_t workspace_bytes, void* stream_) {
    Plan pl;
    TailPlan tp;
    int st = parse(prob, epi, pl, tp, false, true);
    if (st) return st;
    if (!tp.ok) return CNSN_E_UNSUPPORTED;
    if (!x || !z || !workspace) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)z | (uintptr_t)workspace) & 15u) != 0) return CNSN_E_ALIGN;
    if (!gate_ok(g) || !tail_ok(tail)) return CNSN_E_NULL;
    if (saved && !bn_stats) return CNSN_E_NULL;  // (a backward will follow: it needs the statistics)
    if (workspace_bytes < workspace_bytes_of(pl)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    double* saved_d = saved ? (double*)saved : nullptr;
    const MonoArgs ma = make_args(pl, tp.mp);
    TailDev tl{tail->weight, tail->bias, tail->running_mean, tail->running_var, bn_stats, nullptr, nullptr, z, nullptr,
               tail->eps, tail->momentum, tail->training ? 1 : 0, (long long*)tail->num_batches_tracked};
    int status = CNSN_E_UNSUPPORTED;
    dispatch_tail(pl.pr.dtype, tp.mp.vec, tp.mp.lpp, tp.mp.rmax, [&](auto tt, auto vt, auto lt, auto rt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value, RMAX = decltype(rt)::value;
        auto kern = mono_fwd_kernel<T, VEC, LPP, RMAX, true, true>;
        kern<<<mono_grid(pl.pr.C), kMonoBlock, tp.mp.lds, stream>>>(ma, (const T*)x,
                                                                    (const T*)(tp.add == ADD_PRE ? tp.addend : nullptr), (T*)y,
                                                                    gate_dev(g), GateDev{}, saved_d, tp.add, 0, tl);
        const hipError_t e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    });
    return status;
}

int cnsn_backward_bnrelu(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* tail,
                         const void* grad_y, const void* grad_z, const void* x, const cnsn_gate_t* g, const float* saved,
                         const float* bn_stats, void* grad_x, const cnsn_gate_grad_t* dg, float* d_bn_weight,
                         float* d_bn_bias, void* workspace, size_t workspace_bytes, void* stream_) {
    Plan pl;
    TailPlan tp;
    int st = parse(prob, epi, pl, tp, true, true);
    if (st) return st;
    if (!tp.ok) return CNSN_E_UNSUPPORTED;
    if (!grad_z || !x || !grad_x || !saved || !bn_stats || !workspace || !d_bn_weight || !d_bn_bias) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_z | (uintptr_t)grad_x | (uintptr_t)workspace) & 15u) != 0)
        return CNSN_E_ALIGN;
    if (!gate_ok(g) || !gate_grad_ok(dg) || !tail_ok(tail)) return CNSN_E_NULL;
    if (workspace_bytes < workspace_bytes_of(pl)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const MonoArgs ma = make_args(pl, tp.mp);
    TailDev tl{tail->weight, tail->bias, tail->running_mean, tail->running_var, const_cast<float*>(bn_stats), d_bn_weight,
               d_bn_bias, nullptr, grad_z, tail->eps, tail->momentum, tail->training ? 1 : 0, nullptr};
    int status = CNSN_E_UNSUPPORTED;
    dispatch_tail(pl.pr.dtype, tp.mp.vec, tp.mp.lpp, tp.mp.rmax, [&](auto tt, auto vt, auto lt, auto rt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value, RMAX = decltype(rt)::value;
        auto kern = mono_bwd_kernel<T, VEC, LPP, RMAX, true, true>;
        kern<<<mono_grid(pl.pr.C), kMonoBlock, tp.mp.lds, stream>>>(
            ma, (const T*)grad_y, (const T*)x, (const T*)(tp.add == ADD_PRE ? tp.addend : nullptr), (T*)grad_x, gate_dev(g),
            GateDev{}, gate_grad_dev(dg), GateGradDev{}, (const double*)saved, tp.add, 0, tl);
        const hipError_t e = hipGetLastError();
        status = e == hipSuccess ? CNSN_OK : (int)e;
    });
    return status;
}

}  // extern "C"
