// libcnsn_hip.so — host side of the C ABI declared in include/cnsn_hip.h.
// Validates arguments, picks the launch shape (vector width, lanes per plane) and enqueues the
// kernels on the caller's stream.  No allocation, no synchronisation, no global state.
#include "../../include/cnsn_hip.h"
#include "cnsn_env.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cnsn_device.h"
#include "cnsn_host_plan.h"
#include "cnsn_nhwc.h"
#include "cnsn_local.h"
#include "cnsn_mid_kernels.h"
#include "cnsn_mono.h"
#include "cnsn_wide.h"
#include "cnsn_packed.h"
#include "cnsn_resident_fused.h"
#include "cnsn_resident_kernels.h"
#include "cnsn_resident_sn.h"
#include "cnsn_stream_kernels.h"

using namespace cnsn;

namespace cnsn {

// channels per workgroup of the mid kernels: tiles once there are enough channels to fill the chip with tiles (measured at
// C = 2048, N = 256: mid_fwd 115 -> see profiles/r01_small_planes.md) — 8 from C = 2048 on, 4 between 512 and 2048 so that those
// calls still start 128-256 workgroups (profiles/r05_mid_blocks.md); CNSN_MID_TILE=1 disables, =8 restores the single tile size
static int mid_tile(const cnsn_problem_t& p) {
    int only8 = 0;
    if (const char* e = knob(K_MID_TILE)) {
        if (e[0] == '1') return 1;
        only8 = e[0] == '8';
    }
    if (p.C >= 512 && p.C % 8 == 0 && (p.C >= 2048 || only8)) return 8;
    return (p.C >= 512 && p.C % 4 == 0) ? 4 : 1;
}

// threads per workgroup of the channel kernels: 1024 once a thread of a 256-thread workgroup would walk more than two instances
// per sweep (CNSN_MID_BLOCK=256 keeps the small workgroups)
static int mid_block(const cnsn_problem_t& p, int tile) {
    if (const char* e = knob(K_MID_BLOCK))
        if (atoi(e) == 256) return 256;
    return (long long)p.N * tile > 2 * kBlock ? 1024 : kBlock;
}

#define CNSN_MID_DISPATCH(KERNEL, ...)                                                               \
    do {                                                                                             \
        const int tile = mid_tile(pl.pr), blk = mid_block(pl.pr, tile);                              \
        if (tile == 8 && blk == 1024)                                                                \
            KERNEL<8, 1024><<<pl.pr.C / 8, 1024, 0, stream>>>(__VA_ARGS__);                          \
        else if (tile == 8)                                                                          \
            KERNEL<8, kBlock><<<pl.pr.C / 8, kBlock, 0, stream>>>(__VA_ARGS__);                      \
        else if (tile == 4 && blk == 1024)                                                           \
            KERNEL<4, 1024><<<pl.pr.C / 4, 1024, 0, stream>>>(__VA_ARGS__);                          \
        else if (tile == 4)                                                                          \
            KERNEL<4, kBlock><<<pl.pr.C / 4, kBlock, 0, stream>>>(__VA_ARGS__);                      \
        else if (blk == 1024)                                                                        \
            KERNEL<1, 1024><<<pl.pr.C, 1024, 0, stream>>>(__VA_ARGS__);                              \
        else                                                                                         \
            KERNEL<1, kBlock><<<pl.pr.C, kBlock, 0, stream>>>(__VA_ARGS__);                          \
    } while (0)

void launch_mid_fwd(const Plan& pl, const double* mom, const int64_t* perm, const int64_t* chan_perm, GateDev g,
                    GateDev f, float* coef, double* saved, hipStream_t stream) {
    CNSN_MID_DISPATCH(mid_fwd_kernel, pl.mid, mom, perm, chan_perm, g, f, coef, saved);
}

void launch_mid_bwd(const Plan& pl, const float* sums, const double* saved, const int64_t* perm,
                    const int64_t* chan_perm, GateDev g, GateDev f, GateGradDev dg, GateGradDev df, double* tmp,
                    float* coef, hipStream_t stream) {
    CNSN_MID_DISPATCH(mid_bwd_a_kernel, pl.mid, sums, saved, perm, chan_perm, g, f, dg, df, tmp, coef);
    if (pl.mid.cn_active)  // (without CrossNorm mid_bwd_a has written the coefficients itself)
        mid_bwd_b_kernel<<<(int)((pl.P + kBlock - 1) / kBlock), kBlock, 0, stream>>>(pl.mid, saved, tmp, coef);
}
#undef CNSN_MID_DISPATCH

}  // namespace cnsn

extern "C" {

int cnsn_abi_version(void) { return CNSN_ABI_VERSION; }

size_t cnsn_context_bytes(const cnsn_problem_t* prob) {
    Plan pl;
    if (make_plan(prob, pl) != CNSN_OK) return 0;
    const cnsn_problem_t& p = pl.pr;
    if (p.layout == CNSN_LAYOUT_NHWC)  // the single-launch channels-last kernels keep their barrier counter in the control block
        return (nhwc_slim_record(pl) && nhwc_supported(pl, false) && p.N <= kBlock) ? (size_t)kCtlBytes + kBarBlock + 2 * kPongRegion : 0;
    bool any = false;
    for (int bw = 0; bw < 2 && !any; ++bw)
        any = resident_plan(p, pl.boxed, false, bw != 0).ok || resident_fused_plan(p, pl.boxed, false, CNSN_ADD_PRE, bw != 0).ok ||
              resident_split_plan(p, pl.boxed, false, CNSN_ADD_NONE, 0, bw != 0).ok ||
              resident_sn_plan(p, pl.boxed, CNSN_ADD_NONE, 0, bw != 0).ok || resident_sn_plan(p, pl.boxed, CNSN_ADD_PRE, 1, bw != 0).ok;
    // control block + one tagged granule per exchanged scalar (six per plane forward with crop boxes: the most)
    // (+ 256: the scalar-path gather reads whole 256-byte groups)
    if (!any) return 0;
    const size_t general = kCtlBytes + (size_t)p.N * p.C * 6 * 8 + 256, sn = resident_sn_exchange_bytes(p);
    // + the barrier block (resident_bar_area) and the two untagged granule regions at the end (resident_pong_acquire)
    return (general > sn ? general : sn) + kBarBlock + 2 * kPongRegion;
}

int cnsn_context_init(void* context, size_t bytes, void* stream) {
    if (!context) return CNSN_E_NULL;
    if (((uintptr_t)context & 15u) != 0) return CNSN_E_ALIGN;
    if (bytes < (size_t)kCtlBytes) return CNSN_E_WORKSPACE;
    const hipError_t e = hipMemsetAsync(context, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    resident_context_forget(context);
    return CNSN_OK;
}

int cnsn_resident_timeouts(void) { return resident_timeouts(); }
int cnsn_resident_rearm(void) { return resident_rearm(); }
int cnsn_resident_degraded(void) { return resident_degraded() ? 1 : 0; }
void cnsn_resident_enable(int on) { resident_set_enabled(on != 0); }
void cnsn_reload_env(void) { reload_knobs(); }
void cnsn_set_wait_ms(int ms) { resident_set_wait_ms(ms); }
int cnsn_wait_ms(void) { return (int)(resident_wait_ticks() / 100000ll); }
void cnsn_set_headroom_cus(int n) { resident_set_headroom_cus(n); }
int cnsn_headroom_cus(void) { return resident_headroom_cus(); }

const char* cnsn_status_string(int status) {
    switch (status) {
        case CNSN_OK: return "ok";
        case CNSN_E_NULL: return "required pointer is NULL";
        case CNSN_E_SHAPE: return "bad N,C,H,W";
        case CNSN_E_DTYPE: return "unknown dtype";
        case CNSN_E_ALIGN: return "activation pointer not 16-byte aligned";
        case CNSN_E_BOX: return "box outside the plane or empty";
        case CNSN_E_WORKSPACE: return "workspace too small";
        case CNSN_E_BATCH: return "SelfNorm training needs more than one instance per channel";
        case CNSN_E_STRUCT: return "cnsn_problem_t size mismatch";
        case CNSN_E_UNSUPPORTED: return "unsupported request";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
    }
}

size_t cnsn_saved_floats(const cnsn_problem_t* prob) {
    Plan pl;
    if (make_plan(prob, pl) != CNSN_OK) return 0;
    return saved_floats_of(pl);
}

size_t cnsn_workspace_bytes(const cnsn_problem_t* prob) {
    Plan pl;
    if (make_plan(prob, pl) != CNSN_OK) return 0;
    if (pl.pr.layout == CNSN_LAYOUT_NHWC)  // (+ the pixel chunks' partial sums and the plane-order rows: cnsn_nhwc.hip)
        return nhwc_workspace_bytes(pl);
    return workspace_bytes_of(pl);
}

int cnsn_forward(const cnsn_problem_t* prob, const void* x, const int64_t* perm, const int64_t* chan_perm,
                 const cnsn_gate_t* g, const cnsn_gate_t* f, void* y, float* saved, void* workspace,
                 size_t workspace_bytes, void* stream_) {
    if (prob && prob->layout == CNSN_LAYOUT_NHWC)  // channels-last: one entry point for every epilogue (cnsn_fused.hip)
        return cnsn_forward_fused(prob, nullptr, x, perm, chan_perm, g, f, y, saved, workspace, workspace_bytes, stream_);
    Plan pl;
    int st = make_plan(prob, pl);
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

    double* mom = (double*)workspace;
    double* saved_d = saved ? (double*)saved : mom + 6 * pl.P;
    float* coef = (float*)(mom + 6 * pl.P + saved_doubles_of(pl));

    if (resident_sn_prefers(p, pl.boxed, CNSN_ADD_NONE, 0, false)) {
        st = resident_sn_forward(pl.pr, pl.mid, CNSN_ADD_NONE, 0, x, nullptr, gate_dev(g), y, saved ? saved_d : nullptr,
                                 workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    {
        const WidePlan wp = wide_plan(pl, 0, false, chan_perm != nullptr);  // planes that are no whole number of vectors (7x7): channel groups in registers
        if (wp.ok && !perm_inline) {
            st = wide_forward(pl, wp, 0, 0, x, nullptr, perm, gate_dev(g), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_plan(pl, 0, false);  // small planes, SelfNorm alone: the channel in one workgroup's registers
        if (mp.ok) {
            st = mono_forward(pl, mp, 0, 0, x, nullptr, gate_dev(g), gate_dev(f), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_cn_plan(pl, chan_perm != nullptr, 0, false);  // small planes WITH CrossNorm, the same frame
        if (mp.ok && !perm_inline) {
            st = mono_cn_forward(pl, mp, 0, 0, x, nullptr, perm, gate_dev(g), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const LocalPlan lp = local_plan(pl, 0, false);  // small planes, SelfNorm alone: no exchange, no side arrays
        if (lp.ok) {
            st = local_forward(pl, lp, 0, 0, x, nullptr, gate_dev(g), gate_dev(f), y, saved ? saved_d : nullptr, stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    if (resident_sn_plan(p, pl.boxed, CNSN_ADD_NONE, 0, false).ok) {  // SelfNorm alone: partial batch moments exchanged
        st = resident_sn_forward(pl.pr, pl.mid, CNSN_ADD_NONE, 0, x, nullptr, gate_dev(g), y, saved ? saved_d : nullptr,
                                 workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, false).ok) {
        st = resident_pipe_forward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, x, perm, gate_dev(g), gate_dev(f), y,
                                   saved ? saved_d : nullptr, workspace, stream);  // large items: loads under the exchange
        if (st != CNSN_E_UNSUPPORTED) return st;
        st = resident_forward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, x, perm, gate_dev(g), gate_dev(f), y,
                              saved ? saved_d : nullptr, workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;  // otherwise: fall through to the two-pass strategy
    }
    if (resident_split_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, CNSN_ADD_NONE, 0, false).ok) {  // large planes
        st = resident_split_forward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, CNSN_ADD_NONE, 0, x, nullptr, perm, gate_dev(g),
                                    gate_dev(f), y, saved ? saved_d : nullptr, workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }

    if (perm_inline) return CNSN_E_UNSUPPORTED;  // (the two-pass mid kernels read the device array)
    PackedGeom pg;
    if (packed_plan(pl, pg)) {  // small planes: runs of planes staged through LDS (cnsn_packed_kernels.h)
        packed_stats(pl, pg, 0, x, nullptr, mom, stream);
        launch_mid_fwd(pl, mom, perm, chan_perm, gate_dev(g), gate_dev(f), coef, saved_d, stream);
        packed_apply_fwd(pl, pg, 0, 0, x, nullptr, y, coef, stream);
        return launch_status();
    }

    const int blocks = blocks_for(pl.geom.P, pl.shape.lpp);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (pl.boxed)
            plane_stats_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)x, pl.geom, mom, nullptr, 0.f);
        else
            plane_stats_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, pl.geom, mom, nullptr, 0.f);
    });
    launch_mid_fwd(pl, mom, perm, chan_perm, gate_dev(g), gate_dev(f), coef, saved_d, stream);
    const size_t P = pl.P;
    ApplyCoef cf{coef + FC_A_IN * P, coef + FC_XR * P, coef + FC_B_IN * P, coef + FC_A_OUT * P, coef + FC_B_OUT * P};
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (pl.boxed)
            apply_fwd_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)x, (T*)y, pl.geom, cf);
        else
            apply_fwd_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, (T*)y, pl.geom, cf);
    });
    return launch_status();
}

int cnsn_backward(const cnsn_problem_t* prob, const void* grad_y, const void* x, const int64_t* perm,
                  const int64_t* chan_perm, const cnsn_gate_t* g, const cnsn_gate_t* f, const float* saved,
                  void* grad_x, const cnsn_gate_grad_t* dg, const cnsn_gate_grad_t* df, void* workspace,
                  size_t workspace_bytes, void* stream_) {
    if (prob && prob->layout == CNSN_LAYOUT_NHWC)
        return cnsn_backward_fused(prob, nullptr, grad_y, x, perm, chan_perm, g, f, saved, grad_x, nullptr, dg, df, workspace,
                                   workspace_bytes, stream_);
    Plan pl;
    int st = make_plan(prob, pl);
    if (st) return st;
    const cnsn_problem_t& p = pl.pr;
    if (!grad_y || !x || !grad_x || !saved || !workspace) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x | (uintptr_t)workspace) & 15u) != 0)
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

    if (resident_sn_prefers(p, pl.boxed, CNSN_ADD_NONE, 0, true)) {
        st = resident_sn_backward(pl.pr, pl.mid, CNSN_ADD_NONE, 0, grad_y, x, nullptr, gate_dev(g), saved_d, grad_x,
                                  gate_grad_dev(dg), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    {
        const WidePlan wp = wide_plan(pl, 0, true, chan_perm != nullptr);
        if (wp.ok && !perm_inline) {
            st = wide_backward(pl, wp, 0, 0, grad_y, x, nullptr, perm, gate_dev(g), saved_d, grad_x, gate_grad_dev(dg), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_plan(pl, 0, true);
        if (mp.ok) {
            st = mono_backward(pl, mp, 0, 0, grad_y, x, nullptr, gate_dev(g), gate_dev(f), saved_d, grad_x, gate_grad_dev(dg),
                               gate_grad_dev(df), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const MonoPlan mp = mono_cn_plan(pl, chan_perm != nullptr, 0, true);
        if (mp.ok && !perm_inline) {
            st = mono_cn_backward(pl, mp, 0, 0, grad_y, x, nullptr, perm, gate_dev(g), saved_d, grad_x, gate_grad_dev(dg), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    {
        const LocalPlan lp = local_plan(pl, 0, true);
        if (lp.ok) {
            st = local_backward(pl, lp, 0, 0, grad_y, x, nullptr, gate_dev(g), gate_dev(f), saved_d, grad_x, gate_grad_dev(dg),
                                gate_grad_dev(df), stream);
            if (st != CNSN_E_UNSUPPORTED) return st;
        }
    }
    if (resident_sn_plan(p, pl.boxed, CNSN_ADD_NONE, 0, true).ok) {
        st = resident_sn_backward(pl.pr, pl.mid, CNSN_ADD_NONE, 0, grad_y, x, nullptr, gate_dev(g), saved_d, grad_x,
                                  gate_grad_dev(dg), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_sn_cn_plan(p, pl.boxed, chan_perm != nullptr).ok) {  // CrossNorm + SelfNorm: partial batch sums exchanged
        st = resident_sn_cn_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, grad_y, x, perm, gate_dev(g), saved_d, grad_x,
                                     gate_grad_dev(dg), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, true).ok) {
        st = resident_pipe_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, grad_y, x, perm, gate_dev(g), gate_dev(f),
                                    saved_d, grad_x, gate_grad_dev(dg), gate_grad_dev(df), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
        st = resident_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, grad_y, x, perm, gate_dev(g), gate_dev(f),
                               saved_d, grad_x, gate_grad_dev(dg), gate_grad_dev(df), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    if (resident_split_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, CNSN_ADD_NONE, 0, true).ok) {
        st = resident_split_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, CNSN_ADD_NONE, 0, grad_y, x, nullptr, perm,
                                     gate_dev(g), gate_dev(f), saved_d, grad_x, nullptr, gate_grad_dev(dg), gate_grad_dev(df),
                                     workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }

    if (perm_inline) return CNSN_E_UNSUPPORTED;  // (the two-pass mid kernels read the device array)
    PackedGeom pg;
    if (packed_plan(pl, pg)) {
        packed_reduce(pl, pg, 0, 0, grad_y, x, nullptr, saved_d, sums, stream);
        launch_mid_bwd(pl, sums, saved_d, perm, chan_perm, gate_dev(g), gate_dev(f), gate_grad_dev(dg), gate_grad_dev(df),
                       tmp, coef, stream);
        packed_apply_bwd(pl, pg, 0, 0, grad_y, x, nullptr, grad_x, nullptr, coef, saved_d, stream);
        return launch_status();
    }

    const int blocks = blocks_for(pl.geom.P, pl.shape.lpp);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (pl.boxed)
            bwd_reduce_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>(
                (const T*)grad_y, (const T*)x, pl.geom, saved_d + (size_t)SV_MU_C * p.N, saved_d + (size_t)SV_MU_O * p.N, 0, sums);
        else
            bwd_reduce_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>(
                (const T*)grad_y, (const T*)x, pl.geom, saved_d + (size_t)SV_MU_C * p.N, nullptr, 0, sums);
    });
    launch_mid_bwd(pl, sums, saved_d, perm, chan_perm, gate_dev(g), gate_dev(f), gate_grad_dev(dg), gate_grad_dev(df), tmp,
                   coef, stream);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (pl.boxed)
            apply_bwd_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)grad_y, (const T*)x,
                                                                              (T*)grad_x, pl.geom, coef);
        else
            apply_bwd_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)grad_y, (const T*)x,
                                                                               (T*)grad_x, pl.geom, coef);
    });
    return launch_status();
}

int cnsn_plane_stats(const void* x, int dtype, int N, int C, int H, int W, const int32_t* box, float eps,
                     float* mean_std, void* stream_) {
    int st = check_tensor(x, dtype, N, C, H, W);
    if (st) return st;
    if (!mean_std) return CNSN_E_NULL;
    float* mean = mean_std;  // the kernel writes row 0 (mean) and row 1 (std) of the (2, N*C) block
    Box cb;
    bool boxed;
    st = parse_box(box, H, W, cb, boxed);
    if (st) return st;
    const Shape s = pick_shape(dtype, H * W, W, boxed);
    const Geom g = make_geom(N, C, H, W, s.vec, cb, cb);
    const int blocks = blocks_for(g.P, s.lpp);
    hipStream_t stream = (hipStream_t)stream_;
    dispatch(dtype, s, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (boxed)
            plane_stats_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)x, g, nullptr, mean, eps);
        else
            plane_stats_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, g, nullptr, mean, eps);
    });
    return launch_status();
}

int cnsn_plane_stats_backward(const void* x, int dtype, int N, int C, int H, int W, const int32_t* box,
                              const float* mean, const float* std, const float* dmean, const float* dstd, void* dx,
                              void* stream_) {
    int st = check_tensor(x, dtype, N, C, H, W);
    if (st) return st;
    if (!mean || !std || !dmean || !dstd || !dx) return CNSN_E_NULL;
    if (((uintptr_t)dx & 15u) != 0) return CNSN_E_ALIGN;
    Box cb;
    bool boxed;
    st = parse_box(box, H, W, cb, boxed);
    if (st) return st;
    const Shape s = pick_shape(dtype, H * W, W, boxed);
    const Geom g = make_geom(N, C, H, W, s.vec, cb, cb);
    const int blocks = blocks_for(g.P, s.lpp);
    hipStream_t stream = (hipStream_t)stream_;
    dispatch(dtype, s, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (boxed)
            plane_stats_bwd_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>((const T*)x, (T*)dx, g, mean, std,
                                                                                    dmean, dstd);
        else
            plane_stats_bwd_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, (T*)dx, g, mean,
                                                                                     std, dmean, dstd);
    });
    return launch_status();
}

int cnsn_plane_affine(const void* x, int dtype, int N, int C, int H, int W, const float* scale, const float* shift,
                      void* y, void* stream_) {
    int st = check_tensor(x, dtype, N, C, H, W);
    if (st) return st;
    if (!scale || !shift || !y) return CNSN_E_NULL;
    if (((uintptr_t)y & 15u) != 0) return CNSN_E_ALIGN;
    const Shape s = pick_shape(dtype, H * W, W, false);
    const Box whole{0, 0, H, W};
    const Geom g = make_geom(N, C, H, W, s.vec, whole, whole);
    const int blocks = blocks_for(g.P, s.lpp);
    hipStream_t stream = (hipStream_t)stream_;
    ApplyCoef cf{scale, nullptr, shift, nullptr, nullptr};
    dispatch(dtype, s, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        apply_fwd_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)x, (T*)y, g, cf);
    });
    return launch_status();
}

int cnsn_plane_dot(const void* gr, const void* x, int dtype, int N, int C, int H, int W, float* sum_g,
                   void* stream_) {
    int st = check_tensor(x, dtype, N, C, H, W);
    if (st) return st;
    st = check_tensor(gr, dtype, N, C, H, W);
    if (st) return st;
    if (!sum_g) return CNSN_E_NULL;
    const Shape s = pick_shape(dtype, H * W, W, false);
    const Box whole{0, 0, H, W};
    const Geom g = make_geom(N, C, H, W, s.vec, whole, whole);
    const int blocks = blocks_for(g.P, s.lpp);
    hipStream_t stream = (hipStream_t)stream_;
    dispatch(dtype, s, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        bwd_reduce_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)gr, (const T*)x, g, nullptr,
                                                                            nullptr, 0, sum_g);
    });
    return launch_status();
}

int cnsn_plane_dot_shifted(const void* gr, const void* x, int dtype, int N, int C, int H, int W, const double* shift,
                           float* sum_g, void* stream_) {
    int st = check_tensor(x, dtype, N, C, H, W);
    if (st) return st;
    st = check_tensor(gr, dtype, N, C, H, W);
    if (st) return st;
    if (!sum_g || !shift) return CNSN_E_NULL;
    const Shape s = pick_shape(dtype, H * W, W, false);
    const Box whole{0, 0, H, W};
    const Geom g = make_geom(N, C, H, W, s.vec, whole, whole);
    const int blocks = blocks_for(g.P, s.lpp);
    hipStream_t stream = (hipStream_t)stream_;
    dispatch(dtype, s, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        bwd_reduce_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)gr, (const T*)x, g, shift, nullptr, 1,
                                                                            sum_g);
    });
    return launch_status();
}

int cnsn_plane_combine(const void* gr, const void* x, int dtype, int N, int C, int H, int W, const float* coef, void* out,
                       void* stream_) {
    int st = check_tensor(x, dtype, N, C, H, W);
    if (st) return st;
    st = check_tensor(gr, dtype, N, C, H, W);
    if (st) return st;
    if (!coef || !out) return CNSN_E_NULL;
    if (((uintptr_t)out & 15u) != 0) return CNSN_E_ALIGN;
    const Shape s = pick_shape(dtype, H * W, W, false);
    const Box whole{0, 0, H, W};
    const Geom g = make_geom(N, C, H, W, s.vec, whole, whole);
    const int blocks = blocks_for(g.P, s.lpp);
    hipStream_t stream = (hipStream_t)stream_;
    dispatch(dtype, s, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        apply_bwd_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>((const T*)gr, (const T*)x, (T*)out, g, coef);
    });
    return launch_status();
}

}  // extern "C"
