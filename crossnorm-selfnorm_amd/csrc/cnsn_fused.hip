// cnsn_forward_fused / cnsn_backward_fused: the op with the residual block's add and ReLU folded into
// its own launches (include/cnsn_hip.h, "residual-block epilogue").
#include "../../include/cnsn_hip.h"

#include <hip/hip_runtime.h>

#include "cnsn_fused_stream_kernels.h"
#include "cnsn_host_plan.h"
#include "cnsn_local.h"
#include "cnsn_mono.h"
#include "cnsn_nhwc.h"
#include "cnsn_wide.h"
#include "cnsn_packed.h"
#include "cnsn_resident_fused.h"
#include "cnsn_resident_sn.h"

using namespace cnsn;

namespace {

struct EpiPlan {
    int add;  // AddMode
    int relu;
    const void* addend;
    void* sum_out;  // (ABI 8) where x + addend is kept (PRE, channels-last strategies only)
};

int parse_epilogue(const cnsn_epilogue_t* epi, EpiPlan& e) {
    e = EpiPlan{ADD_NONE, 0, nullptr, nullptr};
    if (!epi) return CNSN_OK;
    if (epi->struct_bytes != (int32_t)sizeof(cnsn_epilogue_t)) return CNSN_E_STRUCT;
    if (epi->add_mode != CNSN_ADD_NONE && epi->add_mode != CNSN_ADD_PRE && epi->add_mode != CNSN_ADD_POST)
        return CNSN_E_UNSUPPORTED;
    e.add = epi->add_mode;
    e.relu = epi->relu ? 1 : 0;
    e.addend = epi->addend;
    if (e.add != ADD_NONE) {
        if (!e.addend) return CNSN_E_NULL;
        if (((uintptr_t)e.addend & 15u) != 0) return CNSN_E_ALIGN;
    }
    e.sum_out = epi->sum_out;
    if (e.sum_out) {
        if (e.add != ADD_PRE) return CNSN_E_UNSUPPORTED;
        if (((uintptr_t)e.sum_out & 15u) != 0) return CNSN_E_ALIGN;
    }
    return CNSN_OK;
}

// the same without looking at the addend pointer (cnsn_which_path)
int parse_epilogue_shape(const cnsn_epilogue_t* epi, EpiPlan& e) {
    e = EpiPlan{ADD_NONE, 0, nullptr, nullptr};
    if (!epi) return CNSN_OK;
    if (epi->struct_bytes != (int32_t)sizeof(cnsn_epilogue_t)) return CNSN_E_STRUCT;
    if (epi->add_mode != CNSN_ADD_NONE && epi->add_mode != CNSN_ADD_PRE && epi->add_mode != CNSN_ADD_POST)
        return CNSN_E_UNSUPPORTED;
    e.add = epi->add_mode;
    e.relu = epi->relu ? 1 : 0;
    return CNSN_OK;
}

// call f(IntTag<ADD>) for the runtime add mode
template <typename F>
inline void with_add(int add, F&& f) {
    if (add == ADD_PRE)
        f(IntTag<ADD_PRE>{});
    else if (add == ADD_POST)
        f(IntTag<ADD_POST>{});
    else
        f(IntTag<ADD_NONE>{});
}

}  // namespace

extern "C" {

int cnsn_which_path(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, int has_chan_perm, int backward) {
    EpiPlan e;
    int st = parse_epilogue_shape(epi, e);
    if (st) return st;
    Plan pl;
    st = make_plan(prob, pl);
    if (st) return st;
    const cnsn_problem_t& p = pl.pr;
    const bool chan = p.cn_active && has_chan_perm;
    const bool bwd = backward != 0;
    if (p.layout == CNSN_LAYOUT_NHWC)
        return (chan || !nhwc_supported(pl, false)) ? CNSN_E_UNSUPPORTED : (nhwc_fused_ok(pl) ? CNSN_PATH_RESIDENT : CNSN_PATH_STREAMING);
    // the backward of an epilogue without ReLU and without PRE add is the plain backward
    const bool fused = bwd ? (e.relu || e.add == ADD_PRE) : (e.relu || e.add != ADD_NONE);
    if (resident_sn_prefers(p, pl.boxed, fused ? e.add : ADD_NONE, fused ? e.relu : 0, bwd)) return CNSN_PATH_RESIDENT;
    if (wide_plan(pl, fused ? e.add : 0, bwd, chan).ok) return CNSN_PATH_MONO;  // (channel GROUPS in registers: reported as mono)
    if (mono_plan(pl, fused ? e.add : 0, bwd).ok) return CNSN_PATH_MONO;
    if (mono_cn_plan(pl, chan, fused ? e.add : 0, bwd).ok) return CNSN_PATH_MONO;
    if (local_plan(pl, fused ? e.add : 0, bwd).ok) return CNSN_PATH_LOCAL;
    if (resident_sn_plan(p, pl.boxed, fused ? e.add : ADD_NONE, fused ? e.relu : 0, bwd).ok) return CNSN_PATH_RESIDENT;
    if (bwd && !fused && resident_sn_cn_plan(p, pl.boxed, chan).ok) return CNSN_PATH_RESIDENT;
    if (fused ? resident_fused_plan(p, pl.boxed, chan, e.add, bwd).ok : resident_plan(p, pl.boxed, chan, bwd).ok)
        return CNSN_PATH_RESIDENT;
    if (resident_split_plan(p, pl.boxed, chan, fused ? e.add : ADD_NONE, fused ? e.relu : 0, bwd).ok) return CNSN_PATH_RESIDENT;
    PackedGeom pg;
    return packed_plan(pl, pg) ? CNSN_PATH_PACKED : CNSN_PATH_STREAMING;
}

int cnsn_sn_cluster_plan(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, int backward) {
    EpiPlan e;
    int st = parse_epilogue_shape(epi, e);
    if (st) return st;
    Plan pl;
    st = make_plan(prob, pl);
    if (st) return st;
    const bool bwd = backward != 0;
    const bool fused = bwd ? (e.relu || e.add == ADD_PRE) : (e.relu || e.add != ADD_NONE);
    if (pl.pr.layout == CNSN_LAYOUT_NHWC) return 0;  // (channels-last: the two-pass kernels of cnsn_nhwc.hip)
    // (the small-plane strategies come first in every entry point, unless this family is known to be faster)
    if (resident_sn_prefers(pl.pr, pl.boxed, fused ? e.add : ADD_NONE, fused ? e.relu : 0, bwd)) return 1;
    if (wide_plan(pl, fused ? e.add : 0, bwd).ok || mono_plan(pl, fused ? e.add : 0, bwd).ok ||
        local_plan(pl, fused ? e.add : 0, bwd).ok)
        return 0;
    if (bwd && !fused && resident_sn_cn_plan(pl.pr, pl.boxed, false).ok) return 1;  // (un-boxed CrossNorm in front: same family)
    return resident_sn_plan(pl.pr, pl.boxed, fused ? e.add : ADD_NONE, fused ? e.relu : 0, bwd).ok ? 1 : 0;
}

int cnsn_keeps_sum(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi) {
    EpiPlan e;
    if (!prob || parse_epilogue_shape(epi, e) != CNSN_OK || e.add != ADD_PRE || prob->layout != CNSN_LAYOUT_NHWC) return 0;
    Plan pl;
    if (make_plan(prob, pl) != CNSN_OK) return 0;
    return nhwc_supported(pl, false) ? 1 : 0;
}

// ---- the block's last BatchNorm2d in front of the op (cnsn_nhwc_bnhead_kernels.h) ---------------------------------------------
namespace {
int bn_block_parse(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* bn, Plan& pl, EpiPlan& e, bool shape_only,
                   bool check_health = true) {
    int st = shape_only ? parse_epilogue_shape(epi, e) : parse_epilogue(epi, e);
    if (st) return st;
    st = make_plan(prob, pl);
    if (st) return st;
    if (bn && bn->struct_bytes != (int32_t)sizeof(cnsn_bn_tail_t)) return CNSN_E_STRUCT;
    if (pl.pr.layout != CNSN_LAYOUT_NHWC || e.add != ADD_PRE || e.sum_out || !nhwc_bnhead_ok(pl, check_health) || (bn && !bn->training))
        return CNSN_E_UNSUPPORTED;
    return CNSN_OK;
}
}  // namespace

int cnsn_bn_block_plan(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi) {
    Plan pl;
    EpiPlan e;
    const int st = bn_block_parse(prob, epi, nullptr, pl, e, true);
    return st == CNSN_OK ? 1 : (st == CNSN_E_UNSUPPORTED ? 0 : st);
}

int cnsn_forward_bn_block(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* bn,
                          const cnsn_bn_tail_t* bn_skip, const void* conv_out, const cnsn_gate_t* g, void* y, float* saved, float* bn_stats,
                          void* workspace, size_t workspace_bytes, void* stream_) {
    if (!bn) return CNSN_E_NULL;
    if (bn_skip && (bn_skip->struct_bytes != (int32_t)sizeof(cnsn_bn_tail_t))) return CNSN_E_STRUCT;
    if (bn_skip && (!bn_skip->weight || !bn_skip->bias || !bn_skip->running_mean || !bn_skip->running_var)) return CNSN_E_NULL;
    if (bn_skip && !bn_skip->training) return CNSN_E_UNSUPPORTED;
    Plan pl;
    EpiPlan e;
    const int st = bn_block_parse(prob, epi, bn, pl, e, false);
    if (st) return st;
    if (!conv_out || !y || !workspace || !bn_stats || !bn->weight || !bn->bias || !bn->running_mean || !bn->running_var) return CNSN_E_NULL;
    if ((((uintptr_t)conv_out | (uintptr_t)y | (uintptr_t)workspace | (uintptr_t)bn_stats) & 15u) != 0) return CNSN_E_ALIGN;
    if (!gate_ok(g)) return CNSN_E_NULL;
    return nhwc_bnhead_forward(pl, e.relu, *bn, bn_skip, conv_out, e.addend, gate_dev(g), y, saved, bn_stats, workspace, workspace_bytes,
                               (hipStream_t)stream_);
}

int cnsn_backward_bn_block(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const cnsn_bn_tail_t* bn,
                           const cnsn_bn_tail_t* bn_skip, const void* grad_y, const void* conv_out, const cnsn_gate_t* g, const float* saved,
                           const float* bn_stats, void* grad_conv_out, void* grad_identity, const cnsn_gate_grad_t* dg, float* d_bn_weight,
                           float* d_bn_bias, float* d_bn_skip_weight, float* d_bn_skip_bias, void* workspace, size_t workspace_bytes,
                           void* stream_) {
    if (!bn) return CNSN_E_NULL;
    if (bn_skip && (bn_skip->struct_bytes != (int32_t)sizeof(cnsn_bn_tail_t))) return CNSN_E_STRUCT;
    if (bn_skip && (!bn_skip->weight || !d_bn_skip_weight || !d_bn_skip_bias)) return CNSN_E_NULL;
    Plan pl;
    EpiPlan e;
    const int st = bn_block_parse(prob, epi, bn, pl, e, false, false);
    if (st) return st;
    if (!grad_y || !conv_out || !grad_conv_out || !grad_identity || !saved || !bn_stats || !workspace || !bn->weight) return CNSN_E_NULL;
    if ((((uintptr_t)grad_y | (uintptr_t)conv_out | (uintptr_t)grad_conv_out | (uintptr_t)grad_identity | (uintptr_t)workspace |
          (uintptr_t)bn_stats) & 15u) != 0)
        return CNSN_E_ALIGN;
    if (!gate_ok(g) || !dg || !dg->d_fc_weight || !dg->d_bn_weight || !dg->d_bn_bias) return CNSN_E_NULL;
    return nhwc_bnhead_backward(pl, e.relu, *bn, bn_skip, grad_y, conv_out, e.addend, gate_dev(g), saved, bn_stats, grad_conv_out,
                                grad_identity, gate_grad_dev(dg), d_bn_weight, d_bn_bias, d_bn_skip_weight, d_bn_skip_bias, workspace,
                                workspace_bytes, (hipStream_t)stream_);
}

int cnsn_forward_fused(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const void* x, const int64_t* perm,
                       const int64_t* chan_perm, const cnsn_gate_t* g, const cnsn_gate_t* f, void* y, float* saved,
                       void* workspace, size_t workspace_bytes, void* stream_) {
    EpiPlan e;
    int st = parse_epilogue(epi, e);
    if (st) return st;
    if (prob && prob->layout == CNSN_LAYOUT_NHWC) {  // channels-last tensors: the two-pass kernels of cnsn_nhwc.hip, every epilogue
        Plan pl;
        st = make_plan(prob, pl);
        if (st) return st;
        if (!x || !y || !workspace) return CNSN_E_NULL;
        if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace) & 15u) != 0) return CNSN_E_ALIGN;
        if (chan_perm || !nhwc_supported(pl, false)) return CNSN_E_UNSUPPORTED;
        if (pl.pr.sn_active && !gate_ok(g)) return CNSN_E_NULL;
        if (pl.pr.sn_active && pl.pr.sn_two && !gate_ok(f)) return CNSN_E_NULL;
        return nhwc_forward(pl, e.add, e.relu, x, e.addend, perm, gate_dev(g), gate_dev(f), y, saved, workspace, workspace_bytes,
                            (hipStream_t)stream_, e.sum_out);
    }
    if (e.sum_out) return CNSN_E_UNSUPPORTED;  // (cnsn_keeps_sum() == 0: single-touch strategies read x and the addend once anyway)
    if (e.add == ADD_NONE && !e.relu)
        return cnsn_forward(prob, x, perm, chan_perm, g, f, y, saved, workspace, workspace_bytes, stream_);
    Plan pl;
    st = make_plan(prob, pl);
    if (st) return st;
    const cnsn_problem_t& p = pl.pr;
    if (!x || !y || !workspace) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace) & 15u) != 0) return CNSN_E_ALIGN;
    // (ABI 5) no device array: the permutation travels as a launch argument (cnsn_problem_t.perm_host) — cluster-resident
    // kernels only; every other family below needs the array and is skipped, and the call ends CNSN_E_UNSUPPORTED
    const bool perm_inline = p.cn_active && !perm;
    if (perm_inline && (!p.perm_host || chan_perm || p.N > CNSN_PERM_INLINE_MAX)) return p.perm_host ? CNSN_E_UNSUPPORTED : CNSN_E_NULL;
    if (p.sn_active && !gate_ok(g)) return CNSN_E_NULL;
    if (p.sn_active && p.sn_two && !gate_ok(f)) return CNSN_E_NULL;
    if (workspace_bytes < workspace_bytes_of(pl)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    pl.mid.save_coefs = (e.relu && saved) ? 1 : 0;

    double* mom = (double*)workspace;
    double* saved_d = saved ? (double*)saved : mom + 6 * pl.P;
    float* coef = (float*)(mom + 6 * pl.P + saved_doubles_of(pl));

    if (resident_sn_prefers(p, pl.boxed, e.add, e.relu, false)) {
        st = resident_sn_forward(pl.pr, pl.mid, e.add, e.relu, x, e.addend, gate_dev(g), y, saved ? saved_d : nullptr, workspace,
                                 stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    {
        const WidePlan wp = wide_plan(pl, e.add, false, chan_perm != nullptr);
        if (wp.ok && !perm_inline) {
            st = wide_forward(pl, wp, e.add, e.relu, x, e.addend, perm, gate_dev(g), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_plan(pl, e.add, false);
        if (mp.ok) {
            st = mono_forward(pl, mp, e.add, e.relu, x, e.addend, gate_dev(g), gate_dev(f), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_cn_plan(pl, chan_perm != nullptr, e.add, false);
        if (mp.ok && !perm_inline) {
            st = mono_cn_forward(pl, mp, e.add, e.relu, x, e.addend, perm, gate_dev(g), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const LocalPlan lp = local_plan(pl, e.add, false);
        if (lp.ok) {
            st = local_forward(pl, lp, e.add, e.relu, x, e.addend, gate_dev(g), gate_dev(f), y, saved ? saved_d : nullptr,
                               stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    if (resident_sn_plan(p, pl.boxed, e.add, e.relu, false).ok) {
        st = resident_sn_forward(pl.pr, pl.mid, e.add, e.relu, x, e.addend, gate_dev(g), y, saved ? saved_d : nullptr, workspace,
                                 stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_fused_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, e.add, false).ok) {
        st = resident_fused_forward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, e.add, e.relu, x, e.addend, perm, gate_dev(g),
                                    gate_dev(f), y, saved ? saved_d : nullptr, workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_split_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, e.add, e.relu, false).ok) {  // large planes
        st = resident_split_forward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, e.add, e.relu, x, e.addend, perm, gate_dev(g),
                                    gate_dev(f), y, saved ? saved_d : nullptr, workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }

    if (perm_inline) return CNSN_E_UNSUPPORTED;  // (the two-pass mid kernels read the device array)
    PackedGeom pg;
    if (packed_plan(pl, pg)) {
        packed_stats(pl, pg, e.add, x, e.addend, mom, stream);
        launch_mid_fwd(pl, mom, perm, chan_perm, gate_dev(g), gate_dev(f), coef, saved_d, stream);
        packed_apply_fwd(pl, pg, e.add, e.relu, x, e.addend, y, coef, stream);
        return launch_status();
    }

    const int blocks = blocks_for(pl.geom.P, pl.shape.lpp);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (e.add == ADD_PRE) {
            if (pl.boxed)
                fused_stats_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)x, (const T*)e.addend, pl.geom, mom);
            else
                fused_stats_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, (const T*)e.addend, pl.geom, mom);
        } else {
            if (pl.boxed)
                plane_stats_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)x, pl.geom, mom, nullptr, 0.f);
            else
                plane_stats_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, pl.geom, mom, nullptr, 0.f);
        }
    });
    launch_mid_fwd(pl, mom, perm, chan_perm, gate_dev(g), gate_dev(f), coef, saved_d, stream);
    const size_t P = pl.P;
    ApplyCoef cf{coef + FC_A_IN * P, coef + FC_XR * P, coef + FC_B_IN * P, coef + FC_A_OUT * P, coef + FC_B_OUT * P};
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        with_add(e.add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            if (pl.boxed)
                fused_apply_fwd_kernel<T, VEC, LPP, true, ADD><<<blocks, kBlock, 0, stream>>>(
                    (const T*)x, (const T*)e.addend, (T*)y, pl.geom, cf, e.relu);
            else
                fused_apply_fwd_kernel<T, VEC, LPP, false, ADD><<<blocks, kBlock, 0, stream>>>(
                    (const T*)x, (const T*)e.addend, (T*)y, pl.geom, cf, e.relu);
        });
    });
    return launch_status();
}

int cnsn_backward_fused(const cnsn_problem_t* prob, const cnsn_epilogue_t* epi, const void* grad_y, const void* x,
                        const int64_t* perm, const int64_t* chan_perm, const cnsn_gate_t* g, const cnsn_gate_t* f,
                        const float* saved, void* grad_x, void* grad_addend, const cnsn_gate_grad_t* dg,
                        const cnsn_gate_grad_t* df, void* workspace, size_t workspace_bytes, void* stream_) {
    EpiPlan e;
    int st = parse_epilogue(epi, e);
    if (st) return st;
    if (prob && prob->layout == CNSN_LAYOUT_NHWC) {
        Plan pl;
        st = make_plan(prob, pl);
        if (st) return st;
        if (!grad_y || !x || !grad_x || !saved || !workspace) return CNSN_E_NULL;
        if ((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x | (uintptr_t)grad_addend | (uintptr_t)workspace) & 15u) != 0)
            return CNSN_E_ALIGN;
        if (chan_perm || !nhwc_supported(pl, false)) return CNSN_E_UNSUPPORTED;
        if (pl.pr.sn_active && (!gate_ok(g) || !gate_grad_ok(dg))) return CNSN_E_NULL;
        if (pl.pr.sn_active && pl.pr.sn_two && (!gate_ok(f) || !gate_grad_ok(df))) return CNSN_E_NULL;
        return nhwc_backward(pl, e.add, e.relu, grad_y, x, e.addend, perm, gate_dev(g), gate_dev(f), saved, grad_x, grad_addend,
                             gate_grad_dev(dg), gate_grad_dev(df), workspace, workspace_bytes, (hipStream_t)stream_);
    }
    // without a ReLU the gradient of a POST addend is grad_y itself and nothing else changes
    if (!e.relu && e.add != ADD_PRE)
        return cnsn_backward(prob, grad_y, x, perm, chan_perm, g, f, saved, grad_x, dg, df, workspace, workspace_bytes,
                             stream_);
    Plan pl;
    st = make_plan(prob, pl);
    if (st) return st;
    const cnsn_problem_t& p = pl.pr;
    if (!grad_y || !x || !grad_x || !saved || !workspace) return CNSN_E_NULL;
    if (e.add == ADD_POST && !grad_addend) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x | (uintptr_t)grad_addend | (uintptr_t)workspace) & 15u) != 0)
        return CNSN_E_ALIGN;
    // (ABI 5) no device array: the permutation travels as a launch argument (cnsn_problem_t.perm_host) — cluster-resident
    // kernels only; every other family below needs the array and is skipped, and the call ends CNSN_E_UNSUPPORTED
    const bool perm_inline = p.cn_active && !perm;
    if (perm_inline && (!p.perm_host || chan_perm || p.N > CNSN_PERM_INLINE_MAX)) return p.perm_host ? CNSN_E_UNSUPPORTED : CNSN_E_NULL;
    if (p.sn_active && (!gate_ok(g) || !gate_grad_ok(dg))) return CNSN_E_NULL;
    if (p.sn_active && p.sn_two && (!gate_ok(f) || !gate_grad_ok(df))) return CNSN_E_NULL;
    if (workspace_bytes < workspace_bytes_of(pl)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;

    const size_t P = pl.P;
    double* tmp = (double*)workspace;
    float* sums = (float*)(tmp + BT_ROWS * P);
    float* coef = sums + 4 * P;
    const double* saved_d = (const double*)saved;

    if (resident_sn_prefers(p, pl.boxed, e.add, e.relu, true)) {
        st = resident_sn_backward(pl.pr, pl.mid, e.add, e.relu, grad_y, x, e.addend, gate_dev(g), saved_d, grad_x,
                                  gate_grad_dev(dg), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    {
        const WidePlan wp = wide_plan(pl, e.add, true, chan_perm != nullptr);
        if (wp.ok && !perm_inline) {
            st = wide_backward(pl, wp, e.add, e.relu, grad_y, x, e.addend, perm, gate_dev(g), saved_d, grad_x, gate_grad_dev(dg), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_plan(pl, e.add, true);
        if (mp.ok) {
            st = mono_backward(pl, mp, e.add, e.relu, grad_y, x, e.addend, gate_dev(g), gate_dev(f), saved_d, grad_x,
                               gate_grad_dev(dg), gate_grad_dev(df), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_cn_plan(pl, chan_perm != nullptr, e.add, true);
        if (mp.ok && !perm_inline) {
            st = mono_cn_backward(pl, mp, e.add, e.relu, grad_y, x, e.addend, perm, gate_dev(g), saved_d, grad_x,
                                  gate_grad_dev(dg), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const LocalPlan lp = local_plan(pl, e.add, true);
        if (lp.ok) {
            st = local_backward(pl, lp, e.add, e.relu, grad_y, x, e.addend, gate_dev(g), gate_dev(f), saved_d, grad_x,
                                gate_grad_dev(dg), gate_grad_dev(df), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    if (resident_sn_plan(p, pl.boxed, e.add, e.relu, true).ok) {
        st = resident_sn_backward(pl.pr, pl.mid, e.add, e.relu, grad_y, x, e.addend, gate_dev(g), saved_d, grad_x,
                                  gate_grad_dev(dg), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_fused_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, e.add, true).ok) {
        st = resident_fused_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, e.add, e.relu, grad_y, x, e.addend, perm,
                                     gate_dev(g), gate_dev(f), saved_d, grad_x, grad_addend, gate_grad_dev(dg),
                                     gate_grad_dev(df), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_split_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, e.add, e.relu, true).ok) {
        st = resident_split_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, e.add, e.relu, grad_y, x, e.addend, perm,
                                     gate_dev(g), gate_dev(f), saved_d, grad_x, grad_addend, gate_grad_dev(dg),
                                     gate_grad_dev(df), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }

    if (perm_inline) return CNSN_E_UNSUPPORTED;  // (the two-pass mid kernels read the device array)
    PackedGeom pg;
    if (packed_plan(pl, pg)) {
        packed_reduce(pl, pg, e.add, e.relu, grad_y, x, e.addend, saved_d, sums, stream);
        launch_mid_bwd(pl, sums, saved_d, perm, chan_perm, gate_dev(g), gate_dev(f), gate_grad_dev(dg), gate_grad_dev(df),
                       tmp, coef, stream);
        packed_apply_bwd(pl, pg, e.add, e.relu, grad_y, x, e.addend, grad_x, grad_addend, coef, saved_d, stream);
        return launch_status();
    }

    const int blocks = blocks_for(pl.geom.P, pl.shape.lpp);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        with_add(e.add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            if (pl.boxed)
                fused_bwd_reduce_kernel<T, VEC, LPP, true, ADD><<<blocks, kBlock, 0, stream>>>(
                    (const T*)grad_y, (const T*)x, (const T*)e.addend, pl.geom, saved_d, e.relu, sums);
            else
                fused_bwd_reduce_kernel<T, VEC, LPP, false, ADD><<<blocks, kBlock, 0, stream>>>(
                    (const T*)grad_y, (const T*)x, (const T*)e.addend, pl.geom, saved_d, e.relu, sums);
        });
    });
    launch_mid_bwd(pl, sums, saved_d, perm, chan_perm, gate_dev(g), gate_dev(f), gate_grad_dev(dg), gate_grad_dev(df), tmp,
                   coef, stream);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        with_add(e.add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            if (pl.boxed)
                fused_apply_bwd_kernel<T, VEC, LPP, true, ADD><<<blocks, kBlock, 0, stream>>>(
                    (const T*)grad_y, (const T*)x, (const T*)e.addend, (T*)grad_x, (T*)grad_addend, pl.geom, coef, saved_d,
                    e.relu);
            else
                fused_apply_bwd_kernel<T, VEC, LPP, false, ADD><<<blocks, kBlock, 0, stream>>>(
                    (const T*)grad_y, (const T*)x, (const T*)e.addend, (T*)grad_x, (T*)grad_addend, pl.geom, coef, saved_d,
                    e.relu);
        });
    });
    return launch_status();
}

}  // extern "C"
