// libcnsn_hip.so — host side of the C ABI declared in include/cnsn_hip.h.
// Validates arguments, picks the launch shape (vector width, lanes per plane) and enqueues the
// kernels on the caller's stream.  No allocation, no synchronisation, no global state.
#include "../../include/cnsn_hip.h"

#include <hip/hip_runtime.h>

#include "cnsn_device.h"
#include "cnsn_host_common.h"
#include "cnsn_mid_kernels.h"
#include "cnsn_resident_kernels.h"
#include "cnsn_stream_kernels.h"

using namespace cnsn;

namespace {

struct Shape {
    int vec;  // elements per vector access
    int lpp;  // lanes per plane
};

inline Shape pick_shape(int dtype, int M, int Wd, bool boxed) {
    Shape s;
    s.vec = pick_vec(dtype, boxed ? Wd : M);  // boxed kernels need a vector to stay inside one row
    const int nvec = M / s.vec;
    s.lpp = nvec <= 32 ? 16 : (nvec <= 4096 ? 64 : 256);
    return s;
}

// call f(TypeTag<T>, IntTag<VEC>, IntTag<LPP>) for the runtime (dtype, vec, lpp)
template <typename T, typename F>
inline void dispatch_vl(int vec, int lpp, F&& f) {
    auto with_lpp = [&](auto vtag) {
        switch (lpp) {
            case 16: f(TypeTag<T>{}, vtag, IntTag<16>{}); break;
            case 64: f(TypeTag<T>{}, vtag, IntTag<64>{}); break;
            default: f(TypeTag<T>{}, vtag, IntTag<256>{}); break;
        }
    };
    switch (vec) {
        case 8:
            if constexpr (sizeof(T) == 2) {
                with_lpp(IntTag<8>{});
                break;
            }
            [[fallthrough]];
        case 4: with_lpp(IntTag<4>{}); break;
        case 2: with_lpp(IntTag<2>{}); break;
        default: with_lpp(IntTag<1>{}); break;
    }
}
template <typename F>
inline void dispatch(int dtype, Shape s, F&& f) {
    if (dtype == CNSN_F32)
        dispatch_vl<float>(s.vec, s.lpp, f);
    else if (dtype == CNSN_BF16)
        dispatch_vl<bf16_t>(s.vec, s.lpp, f);
    else
        dispatch_vl<_Float16>(s.vec, s.lpp, f);
}

inline int check_tensor(const void* p, int dtype, int N, int C, int H, int W) {
    if (!p) return CNSN_E_NULL;
    if (dtype != CNSN_F32 && dtype != CNSN_BF16 && dtype != CNSN_F16) return CNSN_E_DTYPE;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return CNSN_E_SHAPE;
    if ((int64_t)N * C > (int64_t)1 << 30 || (int64_t)H * W > (int64_t)1 << 30) return CNSN_E_SHAPE;
    if (((uintptr_t)p & 15u) != 0) return CNSN_E_ALIGN;
    return CNSN_OK;
}

// returns CNSN_OK and fills `b`; a box with x1 < 0 (or NULL) means the whole plane
inline int parse_box(const int32_t* in, int H, int W, Box& b, bool& present) {
    present = in && in[0] >= 0;
    if (!present) {
        b = Box{0, 0, H, W};
        return CNSN_OK;
    }
    b = Box{in[0], in[1], in[2], in[3]};
    if (b.r0 < 0 || b.c0 < 0 || b.r1 > H || b.c1 > W || b.r1 <= b.r0 || b.c1 <= b.c0) return CNSN_E_BOX;
    return CNSN_OK;
}

inline Geom make_geom(int N, int C, int H, int W, int vec, Box cb, Box sb) {
    Geom g;
    g.P = N * C;
    g.M = H * W;
    g.Wd = W;
    g.nvec = g.M / vec;
    g.cb = cb;
    g.sb = sb;
    return g;
}

inline int blocks_for(int P, int lpp) {
    const int ppb = kBlock / lpp;
    return (P + ppb - 1) / ppb;
}

inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CNSN_OK : (int)e;
}

struct Plan {
    cnsn_problem_t pr;
    Box cb, sb;
    bool boxed;
    Shape shape;
    Geom geom;
    MidArgs mid;
    size_t P;
};

int make_plan(const cnsn_problem_t* prob, Plan& pl) {
    if (!prob) return CNSN_E_NULL;
    if (prob->struct_bytes != (int32_t)sizeof(cnsn_problem_t)) return CNSN_E_STRUCT;
    pl.pr = *prob;
    const cnsn_problem_t& p = pl.pr;
    if (p.dtype != CNSN_F32 && p.dtype != CNSN_BF16 && p.dtype != CNSN_F16) return CNSN_E_DTYPE;
    if (p.N <= 0 || p.C <= 0 || p.H <= 0 || p.W <= 0) return CNSN_E_SHAPE;
    if ((int64_t)p.N * p.C > (int64_t)1 << 30 || (int64_t)p.H * p.W > (int64_t)1 << 30) return CNSN_E_SHAPE;
    bool hc = false, hs = false;
    pl.cb = Box{0, 0, p.H, p.W};
    pl.sb = pl.cb;
    if (p.cn_active) {
        int st = parse_box(p.content_box, p.H, p.W, pl.cb, hc);
        if (st) return st;
        st = parse_box(p.style_box, p.H, p.W, pl.sb, hs);
        if (st) return st;
    }
    pl.boxed = hc || hs;
    if (p.sn_active && p.sn_training && p.N < 2) return CNSN_E_BATCH;
    pl.shape = pick_shape(p.dtype, p.H * p.W, p.W, pl.boxed);
    pl.geom = make_geom(p.N, p.C, p.H, p.W, pl.shape.vec, pl.cb, pl.sb);
    pl.P = (size_t)p.N * p.C;
    MidArgs& m = pl.mid;
    m.N = p.N;
    m.C = p.C;
    m.M = p.H * p.W;
    m.Mc = pl.cb.area();
    m.Ms = pl.sb.area();
    m.cn_active = p.cn_active ? 1 : 0;
    m.boxed = pl.boxed ? 1 : 0;
    m.sn_active = p.sn_active ? 1 : 0;
    m.sn_two = (p.sn_active && p.sn_two) ? 1 : 0;
    m.sn_training = p.sn_training ? 1 : 0;
    m.lam = p.cn_active ? p.lam : 0.f;
    m.eps_cn = p.eps_cn;
    m.eps_sn = p.eps_sn;
    m.eps_bn = p.eps_bn;
    m.momentum = p.momentum;
    m.inv_n = 1.0 / (double)p.N;
    m.unbias_n = p.N > 1 ? (double)p.N / ((double)p.N - 1.0) : 1.0;
    return CNSN_OK;
}

// `saved` holds doubles (SV_ROWS rows of P + two rows of C); it is sized in floats for the caller
inline size_t saved_doubles_of(const Plan& pl) { return (size_t)SV_ROWS * pl.P + 2 * (size_t)pl.pr.C; }
inline size_t saved_floats_of(const Plan& pl) { return 2 * saved_doubles_of(pl); }
// workspace (bytes): forward  = moments[6P] f64 | saved fallback f64 | coef[FC_ROWS*P] f32
//                    backward = tmp[BT_ROWS*P] f64 | sums[4P] f32 | coef[BC_ROWS*P] f32
inline size_t workspace_bytes_of(const Plan& pl) {
    const size_t fwd = 8 * (6 * pl.P + saved_doubles_of(pl)) + 4 * (size_t)FC_ROWS * pl.P;
    const size_t bwd = 8 * (size_t)BT_ROWS * pl.P + 4 * (size_t)(4 + BC_ROWS) * pl.P;
    return (fwd > bwd ? fwd : bwd) + 256;
}

inline GateDev gate_dev(const cnsn_gate_t* g) {
    GateDev d{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (g) d = GateDev{g->fc_weight, g->bn_weight, g->bn_bias, g->running_mean, g->running_var};
    return d;
}
inline GateGradDev gate_grad_dev(const cnsn_gate_grad_t* g) {
    GateGradDev d{nullptr, nullptr, nullptr};
    if (g) d = GateGradDev{g->d_fc_weight, g->d_bn_weight, g->d_bn_bias};
    return d;
}
inline bool gate_ok(const cnsn_gate_t* g) {
    return g && g->fc_weight && g->bn_weight && g->bn_bias && g->running_mean && g->running_var;
}
inline bool gate_grad_ok(const cnsn_gate_grad_t* g) { return g && g->d_fc_weight && g->d_bn_weight && g->d_bn_bias; }

}  // namespace

extern "C" {

int cnsn_abi_version(void) { return CNSN_ABI_VERSION; }

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
    return workspace_bytes_of(pl);
}

int cnsn_forward(const cnsn_problem_t* prob, const void* x, const int64_t* perm, const int64_t* chan_perm,
                 const cnsn_gate_t* g, const cnsn_gate_t* f, void* y, float* saved, void* workspace,
                 size_t workspace_bytes, void* stream_) {
    Plan pl;
    int st = make_plan(prob, pl);
    if (st) return st;
    const cnsn_problem_t& p = pl.pr;
    if (!x || !y || !workspace) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace) & 15u) != 0) return CNSN_E_ALIGN;
    if (p.cn_active && !perm) return CNSN_E_NULL;
    if (p.sn_active && !gate_ok(g)) return CNSN_E_NULL;
    if (p.sn_active && p.sn_two && !gate_ok(f)) return CNSN_E_NULL;
    if (workspace_bytes < workspace_bytes_of(pl)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;

    double* mom = (double*)workspace;
    double* saved_d = saved ? (double*)saved : mom + 6 * pl.P;
    float* coef = (float*)(mom + 6 * pl.P + saved_doubles_of(pl));

    if (resident_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, false).ok) {
        st = resident_forward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, x, perm, gate_dev(g), gate_dev(f), y,
                              saved ? saved_d : nullptr, workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;  // otherwise: fall through to the two-pass strategy
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
    mid_fwd_kernel<<<p.C, kBlock, 0, stream>>>(pl.mid, mom, perm, chan_perm, gate_dev(g), gate_dev(f), coef, saved_d);
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
    Plan pl;
    int st = make_plan(prob, pl);
    if (st) return st;
    const cnsn_problem_t& p = pl.pr;
    if (!grad_y || !x || !grad_x || !saved || !workspace) return CNSN_E_NULL;
    if ((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x | (uintptr_t)workspace) & 15u) != 0)
        return CNSN_E_ALIGN;
    if (p.cn_active && !perm) return CNSN_E_NULL;
    if (p.sn_active && (!gate_ok(g) || !gate_grad_ok(dg))) return CNSN_E_NULL;
    if (p.sn_active && p.sn_two && (!gate_ok(f) || !gate_grad_ok(df))) return CNSN_E_NULL;
    if (workspace_bytes < workspace_bytes_of(pl)) return CNSN_E_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;

    const size_t P = pl.P;
    double* tmp = (double*)workspace;
    float* sums = (float*)(tmp + BT_ROWS * P);
    float* coef = sums + 4 * P;
    const double* saved_d = (const double*)saved;

    if (resident_plan(p, pl.boxed, p.cn_active && chan_perm != nullptr, true).ok) {
        st = resident_backward(pl.pr, pl.cb, pl.sb, pl.boxed, pl.mid, grad_y, x, perm, gate_dev(g), gate_dev(f),
                               saved_d, grad_x, gate_grad_dev(dg), gate_grad_dev(df), workspace, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }

    const int blocks = blocks_for(pl.geom.P, pl.shape.lpp);
    dispatch(p.dtype, pl.shape, [&](auto tt, auto vt, auto lt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, LPP = decltype(lt)::value;
        if (pl.boxed)
            bwd_reduce_kernel<T, VEC, LPP, true><<<blocks, kBlock, 0, stream>>>(
                (const T*)grad_y, (const T*)x, pl.geom, saved_d + SV_MU_C, saved_d + SV_MU_O, SV_ROWS, sums);
        else
            bwd_reduce_kernel<T, VEC, LPP, false><<<blocks, kBlock, 0, stream>>>(
                (const T*)grad_y, (const T*)x, pl.geom, saved_d + SV_MU_C, nullptr, SV_ROWS, sums);
    });
    mid_bwd_a_kernel<<<p.C, kBlock, 0, stream>>>(pl.mid, sums, saved_d, perm, chan_perm, gate_dev(g), gate_dev(f),
                                                gate_grad_dev(dg), gate_grad_dev(df), tmp);
    mid_bwd_b_kernel<<<(int)((P + kBlock - 1) / kBlock), kBlock, 0, stream>>>(pl.mid, saved_d, tmp, coef);
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

}  // extern "C"
