// "Mid" kernels of the two-pass strategy: everything the fused op does on N*C scalars between two
// tensor passes.  One workgroup per channel (SelfNorm's BatchNorm1d couples the N planes of a
// channel, models/cnsn.py:121,138); the per-plane algebra lives in cnsn_algebra.h.
#pragma once
#include "cnsn_algebra.h"
#include "cnsn_device.h"
#include "cnsn_layout.h"

namespace cnsn {

// Channel tiling of the mid kernels.  With one workgroup per channel (TC = 1) thread n touches plane n*C + c: every
// access of a wave lands in a different 64-byte sector and 16 workgroups fetch the same sectors — fine while C*N is
// small, the bound of the whole two-pass path when the planes are tiny (C = 2048, 7x7: 250 of 420 us).  With TC > 1 a
// workgroup takes TC adjacent channels: thread t works on channel c0 + t % TC and instances t / TC, t / TC + 256/TC, ...
// so that the TC planes of one instance form one contiguous piece of every side array.
// BLK threads per workgroup: a thread walks N / (BLK / TC) instances in each of the three sweeps, one dependent chain of loads,
// double-precision algebra and stores per instance — at TC = 8, N = 256 and 256 threads that is 24 such steps and the kernel
// takes 32 us whatever C is (profiles/r05_mid_blocks.md); 1024 threads make it 6.
template <int NACC, int TC, int BLK>
__device__ __forceinline__ void tile_sum_d(double (&acc)[NACC], double* lds) {
    if constexpr (TC == 1 && BLK == kBlock) {
        block_sum_d<NACC>(acc, lds);
    } else {  // sum over the threads that share threadIdx.x % TC; lds: (BLK/64) * TC * NACC doubles
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            double v = acc[k];
#pragma unroll
            for (int o = TC; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            acc[k] = v;
        }
        __syncthreads();
        if (lane < TC) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) lds[(wave * TC + lane) * NACC + k] = acc[k];
        }
        __syncthreads();
        const int cl = threadIdx.x % TC;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            double t = 0.0;
            for (int w = 0; w < BLK / 64; ++w) t += lds[(w * TC + cl) * NACC + k];
            acc[k] = t;
        }
    }
}

template <int TC, int BLK>
__global__ __launch_bounds__(BLK) void mid_fwd_kernel(MidArgs a, const double* __restrict__ mom,
                                                         const int64_t* __restrict__ perm,
                                                         const int64_t* __restrict__ chan_perm, GateDev gg, GateDev gf,
                                                         float* __restrict__ coef, double* __restrict__ saved) {
    __shared__ double red[(BLK / 64) * 2 * TC];
    constexpr int NSTEP = BLK / TC;                    // instances per sweep step
    const int n0 = threadIdx.x / TC;                   // first instance of this thread
    const bool lead = n0 == 0;                         // the thread that writes its channel's per-channel results
    const int c = blockIdx.x * TC + threadIdx.x % TC;  // (C is a multiple of TC)
    const size_t P = (size_t)a.N * a.C;
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
    for (int n = n0; n < a.N; n += NSTEP) {
        const size_t p = (size_t)n * a.C + c;
        const SvRec ps = sv_rec(n, c, a.N);
        Moments o;
        o.mu_c = mom[p];
        o.M2c = mom[P + p];
        o.mu_o = a.boxed ? mom[2 * P + p] : 0.0;
        o.M2o = a.boxed ? mom[3 * P + p] : 0.0;
        o.mu_s = a.boxed ? mom[4 * P + p] : o.mu_c;
        o.M2s = a.boxed ? mom[5 * P + p] : o.M2c;
        double mu_sq = 0.0, M2_sq = 0.0;
        if (a.cn_active) {
            const size_t q = (size_t)perm[n] * a.C + cs;  // style source plane (cnsn.py:66-72)
            mu_sq = a.boxed ? mom[4 * P + q] : mom[q];
            M2_sq = a.boxed ? mom[5 * P + q] : mom[P + q];
        }
        const FwdPlane f = fwd_plane<double>(a, o, mu_sq, M2_sq);
        store_fwd_plane(saved, P, ps, f, a.cn_active);
        if (a.sn_active) {
            const double zg = wg0 * f.mu_p + wg1 * f.sig_p;  // Conv1d k=2 groups=C (cnsn.py:137)
            const double zf = wf0 * f.mu_p + wf1 * f.sig_p;
            saved[sv_at(ps, SV_ZH_G)] = zg;  // parked here until normalised in sweep 3
            saved[sv_at(ps, SV_ZH_F)] = zf;
            sz[0] += zg;
            sz[1] += zf;
        }
    }

    double mg = 0, mf = 0, rg = 1, rf = 1;
    if (a.sn_active) {
        if (a.sn_training) {
            // ---- BatchNorm1d batch statistics over N (biased variance for normalising, :138)
            tile_sum_d<2, TC, BLK>(sz, red);
            mg = sz[0] / a.N;
            mf = sz[1] / a.N;
            double sv[2] = {0.0, 0.0};
            for (int n = n0; n < a.N; n += NSTEP) {
                const size_t p = (size_t)n * a.C + c;
                const SvRec ps = sv_rec(n, c, a.N);
                const double dg = saved[sv_at(ps, SV_ZH_G)] - mg;
                const double df = saved[sv_at(ps, SV_ZH_F)] - mf;
                sv[0] += dg * dg;
                sv[1] += df * df;
            }
            tile_sum_d<2, TC, BLK>(sv, red);
            const double vg = sv[0] / a.N, vf = sv[1] / a.N;
            rg = 1.0 / sqrt(vg + (double)a.eps_bn);
            rf = 1.0 / sqrt(vf + (double)a.eps_bn);
            if (lead) {
                const double mom_ = a.momentum, unb = (double)a.N / ((double)a.N - 1.0);
                gg.run_mean[c] = (float)((1.0 - mom_) * gg.run_mean[c] + mom_ * mg);
                gg.run_var[c] = (float)((1.0 - mom_) * gg.run_var[c] + mom_ * vg * unb);
                if (a.sn_two) {
                    gf.run_mean[c] = (float)((1.0 - mom_) * gf.run_mean[c] + mom_ * mf);
                    gf.run_var[c] = (float)((1.0 - mom_) * gf.run_var[c] + mom_ * vf * unb);
                }
                if (c == 0) {
                    bump_batches_tracked(gg.nbt);
                    if (a.sn_two) bump_batches_tracked(gf.nbt);
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
        if (lead) {
            saved[SV_ROWS * P + c] = rg;
            saved[SV_ROWS * P + a.C + c] = rf;
        }
    }

    // ---- sweep 3: gates and the five forward coefficients of every plane of this channel
    const double gam_g = a.sn_active ? (double)gg.gamma[c] : 0.0, bet_g = a.sn_active ? (double)gg.beta[c] : 0.0;
    const double gam_f = a.sn_two ? (double)gf.gamma[c] : 0.0, bet_f = a.sn_two ? (double)gf.beta[c] : 0.0;
    for (int n = n0; n < a.N; n += NSTEP) {
        const size_t p = (size_t)n * a.C + c;
        const SvRec ps = sv_rec(n, c, a.N);
        double g = 1.0, f = 1.0, zhg = 0.0, zhf = 0.0;
        if (a.sn_active) {
            zhg = (saved[sv_at(ps, SV_ZH_G)] - mg) * rg;
            g = sigmoid_d(gam_g * zhg + bet_g);
            if (a.sn_two) {
                zhf = (saved[sv_at(ps, SV_ZH_F)] - mf) * rf;
                f = sigmoid_d(gam_f * zhf + bet_f);
            }
        }
        saved[sv_at(ps, SV_G)] = g;
        saved[sv_at(ps, SV_ZH_G)] = zhg;
        saved[sv_at(ps, SV_F)] = f;
        saved[sv_at(ps, SV_ZH_F)] = zhf;
        FwdPlane fp;
        fp.mu_c = saved[sv_at(ps, SV_MU_C)];
        const CnRowsT<double> cr = load_cn_rows<double>(a, saved, ps, fp.mu_c);
        fp.a1 = cr.a1;
        fp.m_in = cr.m_in;
        fp.mu_p = saved[sv_at(ps, SV_MU_P)];
        const FwdCoefs k = fwd_coefs<double>(a, fp, g, f);
        coef[FC_A_IN * P + p] = k.a_in;
        coef[FC_XR * P + p] = k.xr;
        coef[FC_B_IN * P + p] = k.b_in;
        coef[FC_A_OUT * P + p] = k.a_out;
        coef[FC_B_OUT * P + p] = k.b_out;
        if (a.save_coefs) store_fwd_coefs(saved, ps, k);
    }
}

// per channel: gate / BatchNorm backward, parameter gradients, statistic gradients per plane;
// scatters the style-statistic gradients to the planes that lent their statistics.
template <int TC, int BLK>
__global__ __launch_bounds__(BLK) void mid_bwd_a_kernel(MidArgs a, const float* __restrict__ sums,
                                                           const double* __restrict__ saved,
                                                           const int64_t* __restrict__ perm,
                                                           const int64_t* __restrict__ chan_perm, GateDev gg, GateDev gf,
                                                           GateGradDev dg, GateGradDev df, double* __restrict__ tmp,
                                                           float* __restrict__ coef) {
    __shared__ double red[(BLK / 64) * 4 * TC];
    constexpr int NSTEP = BLK / TC;
    const int n0 = threadIdx.x / TC;
    const bool lead = n0 == 0;
    const int c = blockIdx.x * TC + threadIdx.x % TC;
    const size_t P = (size_t)a.N * a.C;
    const int cs = (a.cn_active && chan_perm) ? (int)chan_perm[c] : c;

    auto sums_of = [&](size_t p, SvRec ps) {
        return fix_sums<double>(a, sums[p], sums[P + p], a.boxed ? sums[2 * P + p] : 0.f, a.boxed ? sums[3 * P + p] : 0.f,
                        saved[sv_at(ps, SV_MU_C)], a.boxed ? saved[sv_at(ps, SV_MU_O)] : 0.0);
    };

    // ---- sweep 1: dL/dgate -> through the sigmoid; batch sums for BatchNorm backward
    double s[4] = {0, 0, 0, 0};  // sum dt_g, sum dt_g*zh_g, sum dt_f, sum dt_f*zh_f
    if (a.sn_active) {
        for (int n = n0; n < a.N; n += NSTEP) {
            const size_t p = (size_t)n * a.C + c;
            const SvRec ps = sv_rec(n, c, a.N);
            double dtg, dtf;
            const CnRowsT<double> cr = load_cn_rows<double>(a, saved, ps, saved[sv_at(ps, SV_MU_C)]);
            gate_dt<double>(a, sums_of(p, ps), cr.a1, cr.m_in, cr.mu_o, saved[sv_at(ps, SV_MU_P)], saved[sv_at(ps, SV_G)],
                            saved[sv_at(ps, SV_F)], dtg, dtf);
            s[0] += dtg;
            s[1] += dtg * saved[sv_at(ps, SV_ZH_G)];
            s[2] += dtf;
            s[3] += dtf * saved[sv_at(ps, SV_ZH_F)];
            tmp[BT_DT_G * P + p] = dtg;
            tmp[BT_DT_F * P + p] = dtf;
        }
        tile_sum_d<4, TC, BLK>(s, red);
        if (lead) {
            dg.dgamma[c] = (float)s[1];
            dg.dbeta[c] = (float)s[0];
            if (a.sn_two) {
                df.dgamma[c] = (float)s[3];
                df.dbeta[c] = (float)s[2];
            }
        }
    }

    // ---- sweep 2: dz, dw, gradient of the plane statistics, CrossNorm statistic gradients
    BnBwd b{};
    b.s_dt_g = s[0];
    b.s_dtz_g = s[1];
    b.s_dt_f = s[2];
    b.s_dtz_f = s[3];
    if (a.sn_active) {
        b.wg0 = gg.w[2 * c];
        b.wg1 = gg.w[2 * c + 1];
        b.kg = (double)gg.gamma[c] * saved[SV_ROWS * P + c];
        if (a.sn_two) {
            b.wf0 = gf.w[2 * c];
            b.wf1 = gf.w[2 * c + 1];
            b.kf = (double)gf.gamma[c] * saved[SV_ROWS * P + a.C + c];
        }
    }
    double sw[4] = {0, 0, 0, 0};  // sum dz_g*mu_p, dz_g*sig_p, dz_f*mu_p, dz_f*sig_p
    for (int n = n0; n < a.N; n += NSTEP) {
        const size_t p = (size_t)n * a.C + c;
        const SvRec ps = sv_rec(n, c, a.N);
        const double mu_p = saved[sv_at(ps, SV_MU_P)], sig_p = saved[sv_at(ps, SV_SIG_P)];
        const CnRowsT<double> cr = load_cn_rows<double>(a, saved, ps, saved[sv_at(ps, SV_MU_C)]);
        const BwdPlane o =
            bwd_plane<double>(a, b, sums_of(p, ps), a.sn_active ? tmp[BT_DT_G * P + p] : 0.0, a.sn_active ? tmp[BT_DT_F * P + p] : 0.0,
                      saved[sv_at(ps, SV_ZH_G)], saved[sv_at(ps, SV_ZH_F)], saved[sv_at(ps, SV_G)], saved[sv_at(ps, SV_F)],
                      cr.aa, cr.a1, cr.m_in, mu_p, sig_p, cr.sig_c, cr.M2c);
        sw[0] += o.dz_g * mu_p;
        sw[1] += o.dz_g * sig_p;
        sw[2] += o.dz_f * mu_p;
        sw[3] += o.dz_f * sig_p;
        if (!a.cn_active) {
            // no plane lends statistics to another one: the coefficients of dx are complete here, mid_bwd_b_kernel is not launched
            // (launch_mid_bwd) and the two scratch rows it would read are not written
            const BwdCoefs k = bwd_coefs<double>(a, o, 0.0, 0.0, saved[sv_at(ps, SV_G)], cr.a1, cr.m_in, mu_p, saved[sv_at(ps, SV_MU_C)],
                                                 cr.sig_c, cr.mu_s, cr.sig_s);
            coef[BC_CG_IN * P + p] = k.cG_in;
            coef[BC_CX_IN * P + p] = k.cX_in;
            coef[BC_XR_IN * P + p] = k.xr_in;
            coef[BC_C0_IN * P + p] = k.c0_in;
            if (a.boxed) {
                coef[BC_CG_OUT * P + p] = k.cG_out;
                coef[BC_CX_OUT * P + p] = k.cX_out;
                coef[BC_XR_OUT * P + p] = k.xr_out;
                coef[BC_C0_OUT * P + p] = k.c0_out;
                coef[BC_ES * P + p] = k.eS;
                coef[BC_XS * P + p] = k.xs;
                coef[BC_E0 * P + p] = k.e0;
            }
        } else {
            tmp[BT_DMU_P * P + p] = o.dmu_p;
            tmp[BT_K * P + p] = o.k;
        }
        if (a.cn_active) {
            tmp[BT_DMU_C * P + p] = o.Dmu_c;
            tmp[BT_DSIG_C * P + p] = o.Dsig_c;
            const size_t q = (size_t)perm[n] * a.C + cs;  // the plane whose statistics were borrowed
            tmp[BT_E_MU * P + q] = o.Emu;
            tmp[BT_E_SIG * P + q] = o.Esig;
        }
    }
    if (a.sn_active) {
        tile_sum_d<4, TC, BLK>(sw, red);
        if (lead) {
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
    const SvRec ps = sv_rec_of_plane(p, a.N, a.C);
    if (p >= P) return;
    BwdPlane o{};
    o.dmu_p = tmp[BT_DMU_P * P + p];
    o.k = tmp[BT_K * P + p];
    double Emu = 0.0, Esig = 0.0;
    if (a.cn_active) {
        o.Dmu_c = tmp[BT_DMU_C * P + p];
        o.Dsig_c = tmp[BT_DSIG_C * P + p];
        Emu = tmp[BT_E_MU * P + p];
        Esig = tmp[BT_E_SIG * P + p];
    }
    const CnRowsT<double> cr = load_cn_rows<double>(a, saved, ps, saved[sv_at(ps, SV_MU_C)]);
    const BwdCoefs k = bwd_coefs<double>(a, o, Emu, Esig, saved[sv_at(ps, SV_G)], cr.a1, cr.m_in, saved[sv_at(ps, SV_MU_P)],
                                         saved[sv_at(ps, SV_MU_C)], cr.sig_c, cr.mu_s, cr.sig_s);
    coef[BC_CG_IN * P + p] = k.cG_in;
    coef[BC_CX_IN * P + p] = k.cX_in;
    coef[BC_XR_IN * P + p] = k.xr_in;
    coef[BC_C0_IN * P + p] = k.c0_in;
    if (a.boxed) {
        coef[BC_CG_OUT * P + p] = k.cG_out;
        coef[BC_CX_OUT * P + p] = k.cX_out;
        coef[BC_XR_OUT * P + p] = k.xr_out;
        coef[BC_C0_OUT * P + p] = k.c0_out;
        coef[BC_ES * P + p] = k.eS;
        coef[BC_XS * P + p] = k.xs;
        coef[BC_E0 * P + p] = k.e0;
    }
}

}  // namespace cnsn
