// "Mid" kernels: everything the fused op does on N*C scalars between two tensor passes.
// One workgroup per channel (SelfNorm's BatchNorm1d couples the N planes of a channel,
// models/cnsn.py:121,138); arithmetic in double — the work is negligible and it keeps the
// coefficient algebra from adding error to what the plane statistics already carry.
//
// The formulas are the ones of oracle/closed_form.py (checked against autograd through the
// op-for-op oracle); names match that file.
#pragma once
#include "cnsn_device.h"

namespace cnsn {

// SoA rows of `saved` (stride P = N*C), followed by two rows of C (BatchNorm rstd of g and f)
enum SavedRow {
    SV_MU_C = 0,   // mean inside the content box (whole plane without one)
    SV_MU_O,       // mean outside the content box
    SV_M2C,        // sum of squared deviations inside the content box
    SV_SIG_C,      // sqrt(var_c + eps_cn)
    SV_MU_S,       // this plane's own style-box mean   (what it lends as a style source)
    SV_SIG_S,      // this plane's own style-box std
    SV_A,          // sig_s[q] / sig_c
    SV_A1,         // lam + (1-lam)*a : slope applied inside the content box
    SV_M_IN,       // mean of the CrossNorm output inside the content box
    SV_MU_P,       // post-CrossNorm whole-plane mean  (SelfNorm's input statistic)
    SV_SIG_P,      // post-CrossNorm whole-plane std, eps_sn
    SV_G,          // gate g
    SV_ZH_G,       // normalised pre-activation of g
    SV_F,          // gate f (two-gate form)
    SV_ZH_F,
    SV_ROWS
};

// rows of the forward coefficient block handed to apply_fwd_kernel
enum FwdCoefRow { FC_A_IN = 0, FC_XR, FC_B_IN, FC_A_OUT, FC_B_OUT, FC_ROWS };

// rows of the backward scratch written by mid_bwd_a and read by mid_bwd_b
enum BwdTmpRow { BT_DT_G = 0, BT_DT_F, BT_DMU_P, BT_K, BT_DMU_C, BT_DSIG_C, BT_E_MU, BT_E_SIG, BT_ROWS };

// rows of the backward coefficient block handed to apply_bwd_kernel
enum BwdCoefRow {
    BC_CG_IN = 0, BC_CX_IN, BC_XR_IN, BC_C0_IN, BC_CG_OUT, BC_CX_OUT, BC_XR_OUT, BC_C0_OUT, BC_ES, BC_XS, BC_E0,
    BC_ROWS
};

struct GateDev {
    const float* w;      // (C,2)
    const float* gamma;  // (C)
    const float* beta;   // (C)
    float* run_mean;     // (C)
    float* run_var;      // (C)
};
struct GateGradDev {
    float* dw;
    float* dgamma;
    float* dbeta;
};

struct MidArgs {
    int N, C, M;
    int Mc, Ms;  // content / style region sizes (M without a box)
    int cn_active, boxed, sn_active, sn_two, sn_training;
    float lam, eps_cn, eps_sn, eps_bn, momentum;
};

__global__ __launch_bounds__(kBlock) void mid_fwd_kernel(MidArgs a, const double* __restrict__ mom,
                                                         const int64_t* __restrict__ perm,
                                                         const int64_t* __restrict__ chan_perm, GateDev gg, GateDev gf,
                                                         float* __restrict__ coef, double* __restrict__ saved) {
    __shared__ double red[(kBlock / 64) * 2];
    const int c = blockIdx.x;
    const size_t P = (size_t)a.N * a.C;
    const double M = a.M, Mc = a.Mc, Mo = a.M - a.Mc;
    const double lam = a.lam;
    const int cs = (a.cn_active && chan_perm) ? (int)chan_perm[c] : c;

    double wg0 = 0, wg1 = 0, wf0 = 0, wf1 = 0;
    if (a.sn_active) {
        wg0 = gg.w[2 * c];
        wg1 = gg.w[2 * c + 1];
        if (a.sn_two) {
            wf0 = gf.w[2 * c];
            wf1 = gf.w[2 * c + 1];
        }
    }

    // ---- sweep 1: CrossNorm algebra per plane, SelfNorm pre-activations z, sum z over the batch
    double sz[2] = {0.0, 0.0};
    for (int n = threadIdx.x; n < a.N; n += kBlock) {
        const size_t p = (size_t)n * a.C + c;
        double mu_c = mom[p], M2c = mom[P + p], mu_o = 0.0, M2o = 0.0, mu_s = mu_c, M2s = M2c;
        if (a.boxed) {
            mu_o = mom[2 * P + p];
            M2o = mom[3 * P + p];
            mu_s = mom[4 * P + p];
            M2s = mom[5 * P + p];
        }
        const double sig_c = sqrt(M2c / (Mc - 1.0) + (double)a.eps_cn);
        const double sig_s = sqrt(M2s / ((double)a.Ms - 1.0) + (double)a.eps_cn);
        double aa = 1.0, a1 = 1.0, m_in = mu_c, mu_p = mu_c, M2p = M2c;
        if (a.cn_active) {
            const size_t q = (size_t)perm[n] * a.C + cs;  // style source plane (cnsn.py:66-72)
            const double mu_sq = a.boxed ? mom[4 * P + q] : mom[q];
            const double M2_sq = a.boxed ? mom[5 * P + q] : mom[P + q];
            const double sig_sq = sqrt(M2_sq / ((double)a.Ms - 1.0) + (double)a.eps_cn);
            aa = sig_sq / sig_c;
            a1 = lam + (1.0 - lam) * aa;
            m_in = lam * mu_c + (1.0 - lam) * mu_sq;
            mu_p = (Mc * m_in + Mo * mu_o) / M;
            M2p = a1 * a1 * M2c + M2o + (m_in - mu_o) * (m_in - mu_o) * Mc * Mo / M;
        }
        const double sig_p = sqrt(M2p / (M - 1.0) + (double)a.eps_sn);
        saved[SV_MU_C * P + p] = mu_c;
        saved[SV_MU_O * P + p] = mu_o;
        saved[SV_M2C * P + p] = M2c;
        saved[SV_SIG_C * P + p] = sig_c;
        saved[SV_MU_S * P + p] = mu_s;
        saved[SV_SIG_S * P + p] = sig_s;
        saved[SV_A * P + p] = aa;
        saved[SV_A1 * P + p] = a1;
        saved[SV_M_IN * P + p] = m_in;
        saved[SV_MU_P * P + p] = mu_p;
        saved[SV_SIG_P * P + p] = sig_p;
        if (a.sn_active) {
            const double zg = wg0 * mu_p + wg1 * sig_p;  // Conv1d k=2 groups=C (cnsn.py:137)
            const double zf = wf0 * mu_p + wf1 * sig_p;
            saved[SV_ZH_G * P + p] = zg;  // parked here until normalised in sweep 3
            saved[SV_ZH_F * P + p] = zf;
            sz[0] += zg;
            sz[1] += zf;
        }
    }

    double mg = 0, mf = 0, rg = 1, rf = 1;
    if (a.sn_active) {
        if (a.sn_training) {
            // ---- BatchNorm1d batch statistics over N (biased variance for normalising, :138)
            block_sum_d<2>(sz, red);
            mg = sz[0] / a.N;
            mf = sz[1] / a.N;
            double sv[2] = {0.0, 0.0};
            for (int n = threadIdx.x; n < a.N; n += kBlock) {
                const size_t p = (size_t)n * a.C + c;
                const double dg = saved[SV_ZH_G * P + p] - mg;
                const double df = saved[SV_ZH_F * P + p] - mf;
                sv[0] += dg * dg;
                sv[1] += df * df;
            }
            block_sum_d<2>(sv, red);
            const double vg = sv[0] / a.N, vf = sv[1] / a.N;
            rg = 1.0 / sqrt(vg + (double)a.eps_bn);
            rf = 1.0 / sqrt(vf + (double)a.eps_bn);
            if (threadIdx.x == 0) {
                const double mom_ = a.momentum, unb = (double)a.N / ((double)a.N - 1.0);
                gg.run_mean[c] = (float)((1.0 - mom_) * gg.run_mean[c] + mom_ * mg);
                gg.run_var[c] = (float)((1.0 - mom_) * gg.run_var[c] + mom_ * vg * unb);
                if (a.sn_two) {
                    gf.run_mean[c] = (float)((1.0 - mom_) * gf.run_mean[c] + mom_ * mf);
                    gf.run_var[c] = (float)((1.0 - mom_) * gf.run_var[c] + mom_ * vf * unb);
                }
            }
        } else {
            mg = gg.run_mean[c];
            rg = 1.0 / sqrt((double)gg.run_var[c] + (double)a.eps_bn);
            if (a.sn_two) {
                mf = gf.run_mean[c];
                rf = 1.0 / sqrt((double)gf.run_var[c] + (double)a.eps_bn);
            }
        }
        if (threadIdx.x == 0) {
            saved[SV_ROWS * P + c] = rg;
            saved[SV_ROWS * P + a.C + c] = rf;
        }
    }

    // ---- sweep 3: gates and the five forward coefficients of every plane of this channel
    const double gam_g = a.sn_active ? (double)gg.gamma[c] : 0.0, bet_g = a.sn_active ? (double)gg.beta[c] : 0.0;
    const double gam_f = a.sn_two ? (double)gf.gamma[c] : 0.0, bet_f = a.sn_two ? (double)gf.beta[c] : 0.0;
    for (int n = threadIdx.x; n < a.N; n += kBlock) {
        const size_t p = (size_t)n * a.C + c;
        double g = 1.0, f = 1.0, zhg = 0.0, zhf = 0.0;
        if (a.sn_active) {
            zhg = (saved[SV_ZH_G * P + p] - mg) * rg;
            g = 1.0 / (1.0 + exp(-(gam_g * zhg + bet_g)));
            if (a.sn_two) {
                zhf = (saved[SV_ZH_F * P + p] - mf) * rf;
                f = 1.0 / (1.0 + exp(-(gam_f * zhf + bet_f)));
            }
        }
        saved[SV_G * P + p] = g;
        saved[SV_ZH_G * P + p] = zhg;
        saved[SV_F * P + p] = f;
        saved[SV_ZH_F * P + p] = zhf;
        const double a1 = saved[SV_A1 * P + p], m_in = saved[SV_M_IN * P + p], mu_p = saved[SV_MU_P * P + p];
        const double shift = a.sn_two ? mu_p * (f - g) : 0.0;  // x*g + mean*(f-g)  (cnsn.py:148)
        if (a.cn_active) {
            // the kernel evaluates A*(x - xr) + B with xr = float(mu_c): fold the rounding of xr into B
            const double mu_c = saved[SV_MU_C * P + p];
            const float xr = (float)mu_c;
            coef[FC_A_IN * P + p] = (float)(g * a1);
            coef[FC_XR * P + p] = xr;
            coef[FC_B_IN * P + p] = (float)(g * m_in + shift + g * a1 * ((double)xr - mu_c));
        } else {  // SelfNorm alone: y = g*x (+ shift), one rounding like the reference's x*g
            coef[FC_A_IN * P + p] = (float)g;
            coef[FC_XR * P + p] = 0.f;
            coef[FC_B_IN * P + p] = (float)shift;
        }
        coef[FC_A_OUT * P + p] = (float)g;
        coef[FC_B_OUT * P + p] = (float)shift;
    }
}

// per channel: gate / BatchNorm backward, parameter gradients, statistic gradients per plane;
// scatters the style-statistic gradients to the planes that lent their statistics.
__global__ __launch_bounds__(kBlock) void mid_bwd_a_kernel(MidArgs a, const float* __restrict__ sums,
                                                           const double* __restrict__ saved,
                                                           const int64_t* __restrict__ perm,
                                                           const int64_t* __restrict__ chan_perm, GateDev gg, GateDev gf,
                                                           GateGradDev dg, GateGradDev df, double* __restrict__ tmp) {
    __shared__ double red[(kBlock / 64) * 4];
    const int c = blockIdx.x;
    const size_t P = (size_t)a.N * a.C;
    const double M = a.M, Mc = a.Mc;
    const double lam = a.lam;
    const int cs = (a.cn_active && chan_perm) ? (int)chan_perm[c] : c;

    // ---- sweep 1: dL/dgate -> through the sigmoid; batch sums for BatchNorm backward
    double s[4] = {0, 0, 0, 0};  // sum dt_g, sum dt_g*zh_g, sum dt_f, sum dt_f*zh_f
    if (a.sn_active) {
        for (int n = threadIdx.x; n < a.N; n += kBlock) {
            const size_t p = (size_t)n * a.C + c;
            const double mu_c = saved[SV_MU_C * P + p], mu_o = saved[SV_MU_O * P + p];
            // bwd_reduce_kernel shifted by float(mu): sum G*(x-mu) = S2 + (float(mu)-mu)*S1
            const double S1in = sums[p], S2in = sums[P + p] + ((double)(float)mu_c - mu_c) * S1in;
            const double S1out = a.boxed ? sums[2 * P + p] : 0.0;
            const double S2out = a.boxed ? sums[3 * P + p] + ((double)(float)mu_o - mu_o) * S1out : 0.0;
            const double a1 = saved[SV_A1 * P + p], m_in = saved[SV_M_IN * P + p];
            const double mu_p = saved[SV_MU_P * P + p];
            const double S1 = S1in + S1out;
            const double GdotU = a1 * S2in + m_in * S1in + S2out + mu_o * S1out;  // sum G*u
            const double g = saved[SV_G * P + p], zhg = saved[SV_ZH_G * P + p];
            const double dgate_g = a.sn_two ? GdotU - mu_p * S1 : GdotU;
            const double dtg = dgate_g * g * (1.0 - g);
            double dtf = 0.0;
            if (a.sn_two) {
                const double f = saved[SV_F * P + p];
                dtf = mu_p * S1 * f * (1.0 - f);
                s[2] += dtf;
                s[3] += dtf * saved[SV_ZH_F * P + p];
            }
            s[0] += dtg;
            s[1] += dtg * zhg;
            tmp[BT_DT_G * P + p] = dtg;
            tmp[BT_DT_F * P + p] = dtf;
        }
        block_sum_d<4>(s, red);
        if (threadIdx.x == 0) {
            dg.dgamma[c] = (float)s[1];
            dg.dbeta[c] = (float)s[0];
            if (a.sn_two) {
                df.dgamma[c] = (float)s[3];
                df.dbeta[c] = (float)s[2];
            }
        }
    }

    // ---- sweep 2: dz, dw, gradient of the plane statistics, CrossNorm statistic gradients
    double wg0 = 0, wg1 = 0, wf0 = 0, wf1 = 0, kg = 0, kf = 0;
    if (a.sn_active) {
        wg0 = gg.w[2 * c];
        wg1 = gg.w[2 * c + 1];
        kg = (double)gg.gamma[c] * saved[SV_ROWS * P + c];
        if (a.sn_two) {
            wf0 = gf.w[2 * c];
            wf1 = gf.w[2 * c + 1];
            kf = (double)gf.gamma[c] * saved[SV_ROWS * P + a.C + c];
        }
    }
    const double invN = 1.0 / a.N;
    double sw[4] = {0, 0, 0, 0};  // sum dz_g*mu_p, dz_g*sig_p, dz_f*mu_p, dz_f*sig_p
    for (int n = threadIdx.x; n < a.N; n += kBlock) {
        const size_t p = (size_t)n * a.C + c;
        const double mu_c = saved[SV_MU_C * P + p];
        const double S1in = sums[p], S2in = sums[P + p] + ((double)(float)mu_c - mu_c) * S1in;
        const double S1out = a.boxed ? sums[2 * P + p] : 0.0;
        const double a1 = saved[SV_A1 * P + p], m_in = saved[SV_M_IN * P + p];
        const double mu_p = saved[SV_MU_P * P + p], sig_p = saved[SV_SIG_P * P + p];
        const double g = saved[SV_G * P + p];
        double dmu_p = 0.0, dsig_p = 0.0;
        if (a.sn_active) {
            const double dtg = tmp[BT_DT_G * P + p], zhg = saved[SV_ZH_G * P + p];
            const double dzg = kg * (a.sn_training ? dtg - s[0] * invN - zhg * s[1] * invN : dtg);
            dmu_p += dzg * wg0;
            dsig_p += dzg * wg1;
            sw[0] += dzg * mu_p;
            sw[1] += dzg * sig_p;
            if (a.sn_two) {
                const double f = saved[SV_F * P + p];
                const double dtf = tmp[BT_DT_F * P + p], zhf = saved[SV_ZH_F * P + p];
                const double dzf = kf * (a.sn_training ? dtf - s[2] * invN - zhf * s[3] * invN : dtf);
                dmu_p += dzf * wf0 + (f - g) * (S1in + S1out);
                dsig_p += dzf * wf1;
                sw[2] += dzf * mu_p;
                sw[3] += dzf * sig_p;
            }
        }
        const double k = a.sn_active ? dsig_p / (sig_p * (M - 1.0)) : 0.0;
        tmp[BT_DMU_P * P + p] = dmu_p;
        tmp[BT_K * P + p] = k;
        if (a.cn_active) {
            const double aa = saved[SV_A * P + p], sig_c = saved[SV_SIG_C * P + p], M2c = saved[SV_M2C * P + p];
            const double T1 = g * S1in + Mc * dmu_p / M + k * Mc * (m_in - mu_p);
            const double T2 = g * S2in + k * a1 * M2c;
            const double d_a = (1.0 - lam) * T2;
            tmp[BT_DMU_C * P + p] = (-(1.0 - lam) * aa * T1);
            tmp[BT_DSIG_C * P + p] = (-d_a * aa / sig_c);
            const size_t q = (size_t)perm[n] * a.C + cs;  // the plane whose statistics were borrowed
            tmp[BT_E_MU * P + q] = ((1.0 - lam) * T1);
            tmp[BT_E_SIG * P + q] = (d_a / sig_c);
        }
    }
    if (a.sn_active) {
        block_sum_d<4>(sw, red);
        if (threadIdx.x == 0) {
            dg.dw[2 * c] = (float)sw[0];
            dg.dw[2 * c + 1] = (float)sw[1];
            if (a.sn_two) {
                df.dw[2 * c] = (float)sw[2];
                df.dw[2 * c + 1] = (float)sw[3];
            }
        }
    }
}

// per plane: assemble the coefficients of dx = cG*G + cX*(x-xr) + c0 (+ style term)
__global__ __launch_bounds__(kBlock) void mid_bwd_b_kernel(MidArgs a, const double* __restrict__ saved,
                                                           const double* __restrict__ tmp, float* __restrict__ coef) {
    const size_t P = (size_t)a.N * a.C;
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const double M = a.M, Mc = a.Mc, Ms = a.Ms;
    const double a1 = saved[SV_A1 * P + p], m_in = saved[SV_M_IN * P + p], mu_p = saved[SV_MU_P * P + p];
    const double g = saved[SV_G * P + p], mu_c = saved[SV_MU_C * P + p];
    const double dmu_p = tmp[BT_DMU_P * P + p], k = tmp[BT_K * P + p];
    double cX_in = a1 * a1 * k, c0_in = a1 * (dmu_p / M + k * (m_in - mu_p));
    double eS = 0.0, e0 = 0.0;
    if (a.cn_active) {
        const double sig_c = saved[SV_SIG_C * P + p], sig_s = saved[SV_SIG_S * P + p];
        cX_in += tmp[BT_DSIG_C * P + p] / (sig_c * (Mc - 1.0));
        c0_in += tmp[BT_DMU_C * P + p] / Mc;
        eS = tmp[BT_E_SIG * P + p] / (sig_s * (Ms - 1.0));
        e0 = tmp[BT_E_MU * P + p] / Ms;
    }
    if (!a.boxed) {  // style region == content region == plane, mu_s == mu_c: one affine map
        cX_in += eS;
        c0_in += e0;
    }
    // the kernel evaluates c*(x - float(ref)) + c0: fold the rounding of each reference point into c0
    const float xr_in = (float)mu_c;
    coef[BC_CG_IN * P + p] = (float)(a1 * g);
    coef[BC_CX_IN * P + p] = (float)cX_in;
    coef[BC_XR_IN * P + p] = xr_in;
    coef[BC_C0_IN * P + p] = (float)(c0_in + cX_in * ((double)xr_in - mu_c));
    if (a.boxed) {
        const double mu_s = saved[SV_MU_S * P + p];
        const float xr_out = (float)mu_p, xs = (float)mu_s;
        coef[BC_CG_OUT * P + p] = (float)g;
        coef[BC_CX_OUT * P + p] = (float)k;
        coef[BC_XR_OUT * P + p] = xr_out;
        coef[BC_C0_OUT * P + p] = (float)(dmu_p / M + k * ((double)xr_out - mu_p));
        coef[BC_ES * P + p] = (float)eS;
        coef[BC_XS * P + p] = xs;
        coef[BC_E0 * P + p] = (float)(e0 + eS * ((double)xs - mu_s));
    }
}

}  // namespace cnsn
