// Host-side planning shared by the translation units of libcnsn_hip.so: argument validation, launch shape
// (vector width, lanes per plane), the layout of `saved` and of the workspace, type dispatch.
#pragma once
#include <cstdlib>
#include "cnsn_env.h"
#include "../../include/cnsn_hip.h"

#include <hip/hip_runtime.h>

#include "cnsn_algebra.h"
#include "cnsn_device.h"
#include "cnsn_host_common.h"
#include "cnsn_layout.h"

namespace cnsn {

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
    g.N = N;
    g.C = C;
    g.M = H * W;
    g.Wd = W;
    g.nvec = g.M / vec;
    g.cb = cb;
    g.sb = sb;
    g.keep = 0;
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

inline int make_plan(const cnsn_problem_t* prob, Plan& pl) {
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
    // two-pass strategy: tensors of up to 512 MiB read their first pass with the default cache policy — the second pass
    // then hits L2 / the 256 MB Infinity Cache for part of them (measured: (768,3,224,224) bf16 0.362 -> 0.339 ms,
    // (256,3,224,224) fp32 0.237 -> 0.210; at 822 MB non-temporal is 11 % faster: profiles/r01_resident_tuning.md)
    pl.geom.keep = ((size_t)p.N * p.C * p.H * p.W * elem_bytes(p.dtype) <= ((size_t)512 << 20)) ? 1 : 0;
    if (const char* e = knob(K_KEEP)) pl.geom.keep = e[0] == '1' ? 1 : (e[0] == '0' ? 0 : pl.geom.keep);
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
    m.save_coefs = 0;
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
    GateDev d{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (g) d = GateDev{g->fc_weight, g->bn_weight, g->bn_bias, g->running_mean, g->running_var, (long long*)g->num_batches_tracked};
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


// the mid kernels live in cnsn_abi.hip (non-template __global__ functions: one definition); other units launch
// them through these
void launch_mid_fwd(const Plan& pl, const double* mom, const int64_t* perm, const int64_t* chan_perm, GateDev g,
                    GateDev f, float* coef, double* saved, hipStream_t stream);
void launch_mid_bwd(const Plan& pl, const float* sums, const double* saved, const int64_t* perm,
                    const int64_t* chan_perm, GateDev g, GateDev f, GateGradDev dg, GateGradDev df, double* tmp,
                    float* coef, hipStream_t stream);

}  // namespace cnsn
