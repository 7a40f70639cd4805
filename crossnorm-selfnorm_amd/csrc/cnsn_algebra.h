// Per-plane scalar algebra of the fused op, shared by the two-pass "mid" kernels and the
// channel-resident kernels; formulas and names are those of oracle/closed_form.py (checked against
// autograd through the op-for-op oracle).  Templated on the scalar type R:
//   R = double  two-pass mid kernels (N*C scalars in their own launches: precision is free there)
//   R = float   resident kernels (the algebra runs while whole planes sit in VGPRs: registers and
//               latency matter; float is what the reference itself computes these scalars in).
// Whatever R is, the cross-batch reductions and the BatchNorm1d normalisation around this algebra
// are done in double by the callers (they are cancellation-prone, see bwd_plane).
#pragma once
#include "cnsn_device.h"

#define CNSN_ALGEBRA_FN template <typename R> __device__ __forceinline__

namespace cnsn {

struct MidArgs {
    int N, C, M;
    int Mc, Ms;  // content / style region sizes (M without a box)
    int cn_active, boxed, sn_active, sn_two, sn_training;
    float lam, eps_cn, eps_sn, eps_bn, momentum;
    double inv_n;     // 1/N         (host-computed: no fp64 divisions on the device)
    double unbias_n;  // N/(N-1)     (running_var takes the unbiased batch variance)
    int save_coefs;   // forward: also keep the five apply coefficients in `saved` (ReLU-fused backward)
};

struct GateDev {
    const float* w;      // (C,2)  Conv1d(C,C,k=2,groups=C) weight (cnsn.py:119)
    const float* gamma;  // (C)    BatchNorm1d weight
    const float* beta;   // (C)    BatchNorm1d bias
    float* run_mean;     // (C)
    float* run_var;      // (C)
    long long* nbt;      // BatchNorm1d.num_batches_tracked (int64 scalar) or null: the forward adds 1 in training mode
};
// nn.BatchNorm1d.forward's `self.num_batches_tracked.add_(1)` (models/cnsn.py:121,138 call the module): done by the ONE thread
// of the launch that updates channel 0's running statistics
__device__ __forceinline__ void bump_batches_tracked(long long* nbt) {
    if (nbt) *nbt += 1;
}
struct GateGradDev {
    float* dw;
    float* dgamma;
    float* dbeta;
};

// region moments of one plane as pass A produces them
template <typename R>
struct MomentsT {
    R mu_c, M2c;  // inside the content box (whole plane without one)
    R mu_o, M2o;  // outside the content box
    R mu_s, M2s;  // inside the style box
};
using Moments = MomentsT<double>;

// forward per-plane state (rows SV_MU_C .. SV_SIG_P of `saved`)
template <typename R>
struct FwdPlaneT {
    R mu_c, mu_o, M2c, sig_c, mu_s, sig_s, aa, a1, m_in, mu_p, sig_p;
};
using FwdPlane = FwdPlaneT<double>;

// double: IEEE library routines.  float (resident kernels): the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 —
// a handful of instructions and no temporaries where the correctly rounded sequences need a dozen.
__device__ __forceinline__ double sqrt_r(double v) { return sqrt(v); }
__device__ __forceinline__ float sqrt_r(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ double exp_r(double v) { return exp(v); }
__device__ __forceinline__ float exp_r(float v) { return __expf(v); }
__device__ __forceinline__ double div_r(double a, double b) { return a / b; }
__device__ __forceinline__ float div_r(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// CrossNorm algebra of one plane: own moments + the style source's style-box moments
// (cnsn.py:24-29 folded with the box paste :75-82 and the lam blend :87), then the post-CrossNorm
// whole-plane moments by Chan's merge, which is what SelfNorm's calc_ins_mean_std (:133) would see.
CNSN_ALGEBRA_FN FwdPlaneT<R> fwd_plane(const MidArgs& a, const MomentsT<R>& o, R mu_sq, R M2_sq) {
    const R M = a.M, Mc = a.Mc, Mo = a.M - a.Mc, lam = a.lam;
    FwdPlaneT<R> p;
    p.mu_c = o.mu_c;
    p.mu_o = o.mu_o;
    p.M2c = o.M2c;
    p.mu_s = o.mu_s;
    p.sig_c = sqrt_r(div_r(o.M2c, Mc - R(1)) + (R)a.eps_cn);
    p.sig_s = sqrt_r(div_r(o.M2s, (R)a.Ms - R(1)) + (R)a.eps_cn);
    p.aa = R(1);
    p.a1 = R(1);
    p.m_in = o.mu_c;
    p.mu_p = o.mu_c;
    R M2p = o.M2c;
    if (a.cn_active) {
        const R sig_sq = sqrt_r(div_r(M2_sq, (R)a.Ms - R(1)) + (R)a.eps_cn);
        p.aa = div_r(sig_sq, p.sig_c);
        p.a1 = lam + (R(1) - lam) * p.aa;
        p.m_in = lam * o.mu_c + (R(1) - lam) * mu_sq;
        p.mu_p = div_r(Mc * p.m_in + Mo * o.mu_o, M);
        M2p = p.a1 * p.a1 * o.M2c + o.M2o + div_r((p.m_in - o.mu_o) * (p.m_in - o.mu_o) * Mc * Mo, M);
    }
    p.sig_p = sqrt_r(div_r(M2p, M - R(1)) + (R)a.eps_sn);
    return p;
}

CNSN_ALGEBRA_FN R sigmoid_r(R t) { return div_r(R(1), R(1) + exp_r(-t)); }
__device__ __forceinline__ double sigmoid_d(double t) { return sigmoid_r<double>(t); }

struct FwdCoefs {
    float a_in, xr, b_in, a_out, b_out;
};

// y = a_in*(x - xr) + b_in inside the content box, a_out*x + b_out outside
CNSN_ALGEBRA_FN FwdCoefs fwd_coefs(const MidArgs& a, const FwdPlaneT<R>& p, R g, R f) {
    FwdCoefs c;
    const R shift = a.sn_two ? p.mu_p * (f - g) : R(0);  // x*g + mean*(f-g)  (cnsn.py:148)
    if (a.cn_active) {
        // evaluated as A*(x - xr) + B with xr = float(mu_c): the rounding of xr is folded into B
        c.xr = (float)p.mu_c;
        c.a_in = (float)(g * p.a1);
        c.b_in = (float)(g * p.m_in + shift + g * p.a1 * ((R)c.xr - p.mu_c));
    } else {  // SelfNorm alone: y = g*x (+ shift), one rounding like the reference's x*g (:150)
        c.xr = 0.f;
        c.a_in = (float)g;
        c.b_in = (float)shift;
    }
    c.a_out = (float)g;
    c.b_out = (float)shift;
    return c;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <typename R>
struct BwdSumsT {
    R S1in, S2in, S1out, S2out;  // sum G, sum G*(x-mu_c) in the box; sum G, sum G*(x-mu_o) outside
};
using BwdSums = BwdSumsT<double>;

// pass A' shifts x by float(mu): sum G*(x-mu) = S2 + (float(mu)-mu)*S1
CNSN_ALGEBRA_FN BwdSumsT<R> fix_sums(const MidArgs& a, float s1i, float s2i, float s1o, float s2o, double mu_c,
                               double mu_o) {
    BwdSumsT<R> s;
    s.S1in = s1i;
    s.S2in = (R)((double)s2i + ((double)(float)mu_c - mu_c) * (double)s1i);
    s.S1out = a.boxed ? (R)s1o : R(0);
    s.S2out = a.boxed ? (R)((double)s2o + ((double)(float)mu_o - mu_o) * (double)s1o) : R(0);
    return s;
}

// dL/d(pre-sigmoid) of both gates for one plane
CNSN_ALGEBRA_FN void gate_dt(const MidArgs& a, const BwdSumsT<R>& s, R a1, R m_in, R mu_o, R mu_p, R g, R f, R& dtg,
                             R& dtf) {
    const R S1 = s.S1in + s.S1out;
    const R GdotU = a1 * s.S2in + m_in * s.S1in + s.S2out + mu_o * s.S1out;  // sum G*u
    const R dgate_g = a.sn_two ? GdotU - mu_p * S1 : GdotU;
    dtg = dgate_g * g * (R(1) - g);
    dtf = a.sn_two ? mu_p * S1 * f * (R(1) - f) : R(0);
}

// per-channel constants of the BatchNorm1d backward
struct BnBwd {
    double kg, kf;                       // gamma * rstd of each gate
    double s_dt_g, s_dtz_g, s_dt_f, s_dtz_f;  // batch sums of dt and dt*zh
    double wg0, wg1, wf0, wf1;           // Conv1d taps
};

template <typename R>
struct BwdPlaneT {
    R dz_g, dz_f;
    R dmu_p, k;          // gradient wrt the post-CN plane mean; dsig_p / (sig_p*(M-1))
    R Dmu_c, Dsig_c;     // gradient wrt this plane's content statistics
    R Emu, Esig;         // gradient this plane sends to ITS style source's (mean, std)
};
using BwdPlane = BwdPlaneT<double>;

// dz = gamma*rstd*(dt - mean(dt) - zh*mean(dt*zh)) cancels catastrophically when the batch is small
// or the gate saturated: that line is always evaluated in double, whatever R is.
CNSN_ALGEBRA_FN BwdPlaneT<R> bwd_plane(const MidArgs& a, const BnBwd& b, const BwdSumsT<R>& s, double dtg, double dtf,
                                       double zhg, double zhf, R g, R f, R aa, R a1, R m_in, R mu_p, R sig_p, R sig_c,
                                       R M2c) {
    const R M = a.M, Mc = a.Mc, lam = a.lam;
    const double invN = a.inv_n;
    BwdPlaneT<R> o;
    o.dz_g = o.dz_f = R(0);
    R dmu_p = R(0), dsig_p = R(0);
    if (a.sn_active) {
        o.dz_g = (R)(b.kg * (a.sn_training ? dtg - b.s_dt_g * invN - zhg * b.s_dtz_g * invN : dtg));
        dmu_p += o.dz_g * (R)b.wg0;
        dsig_p += o.dz_g * (R)b.wg1;
        if (a.sn_two) {
            o.dz_f = (R)(b.kf * (a.sn_training ? dtf - b.s_dt_f * invN - zhf * b.s_dtz_f * invN : dtf));
            dmu_p += o.dz_f * (R)b.wf0 + (f - g) * (s.S1in + s.S1out);
            dsig_p += o.dz_f * (R)b.wf1;
        }
    }
    o.dmu_p = dmu_p;
    o.k = a.sn_active ? div_r(dsig_p, sig_p * (M - R(1))) : R(0);
    o.Dmu_c = o.Dsig_c = o.Emu = o.Esig = R(0);
    if (a.cn_active) {
        const R T1 = g * s.S1in + div_r(Mc * dmu_p, M) + o.k * Mc * (m_in - mu_p);
        const R T2 = g * s.S2in + o.k * a1 * M2c;
        const R d_a = (R(1) - lam) * T2;
        o.Dmu_c = -(R(1) - lam) * aa * T1;
        o.Dsig_c = -div_r(d_a * aa, sig_c);
        o.Emu = (R(1) - lam) * T1;
        o.Esig = div_r(d_a, sig_c);
    }
    return o;
}

struct BwdCoefs {
    float cG_in, cX_in, xr_in, c0_in, cG_out, cX_out, xr_out, c0_out, eS, xs, e0;
};

// dx = cG*G + cX*(x - xr) + c0 by region, + eS*(x - xs) + e0 inside the style box.
// (Emu_in, Esig_in) = what the plane that borrowed THIS plane's statistics sends back.
CNSN_ALGEBRA_FN BwdCoefs bwd_coefs(const MidArgs& a, const BwdPlaneT<R>& o, R Emu_in, R Esig_in, R g, R a1, R m_in,
                                   R mu_p, double mu_c, R sig_c, double mu_s, R sig_s) {
    const R M = a.M, Mc = a.Mc, Ms = a.Ms;
    R cX_in = a1 * a1 * o.k, c0_in = a1 * (div_r(o.dmu_p, M) + o.k * (m_in - mu_p));
    R eS = R(0), e0 = R(0);
    if (a.cn_active) {
        cX_in += div_r(o.Dsig_c, sig_c * (Mc - R(1)));
        c0_in += div_r(o.Dmu_c, Mc);
        eS = div_r(Esig_in, sig_s * (Ms - R(1)));
        e0 = div_r(Emu_in, Ms);
    }
    if (!a.boxed) {  // style region == content region == plane and mu_s == mu_c: one affine map
        cX_in += eS;
        c0_in += e0;
    }
    // evaluated as c*(x - float(ref)) + c0: the rounding of each reference point is folded into c0
    BwdCoefs c;
    c.xr_in = (float)mu_c;
    c.cG_in = (float)(a1 * g);
    c.cX_in = (float)cX_in;
    c.c0_in = (float)(c0_in + cX_in * (R)((double)c.xr_in - mu_c));
    c.xr_out = (float)mu_p;
    c.xs = (float)mu_s;
    c.cG_out = (float)g;
    c.cX_out = (float)o.k;
    c.c0_out = (float)(div_r(o.dmu_p, M) + o.k * ((R)c.xr_out - mu_p));
    c.eS = (float)eS;
    c.e0 = (float)(e0 + eS * (R)((double)c.xs - mu_s));
    return c;
}

}  // namespace cnsn
