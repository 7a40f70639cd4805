// Channel-in-registers kernels WITH CrossNorm (cn_op_2ins_space_chan, models/cnsn.py:58-91, optionally followed by
// SelfNorm): the small-plane sites where CrossNorm is armed — WideResNet's 16x16 / 8x8 stages with crop boxes
// (models/cifar/wideresnet_cnsn.py:93-96), ResNet's 14x14 / 7x7 stages — ran the packed two-pass kernels at 0.9-2 TB/s in
// round 1 (the cluster kernels need a vector to stay inside one row of the plane; the channel-local and the SelfNorm-only
// mono kernels know no CrossNorm).
//
// Same frame as cnsn_mono_kernels.h: one 1024-thread workgroup holds every plane of a channel in registers.  What
// CrossNorm adds stays inside the workgroup, because the permutation only pairs planes of ONE channel (cnsn.py:62-68):
//   * statistics of three regions per plane — inside the content box, outside it, inside the style box — from one
//     bit mask per lane (every slot row is a whole plane: the masks are the same for all rows; vectors may straddle rows);
//   * the style source's moments are read from the workgroup's LDS (`st[..][perm[n]]`);
//   * backward: the statistic gradients a plane sends to its style source (Emu, Esig) go through LDS and come back by
//     the inverse permutation; dx is piecewise affine over the regions (11 coefficients per plane).
// Algebra: cnsn_algebra.h (fwd_plane / fwd_coefs / gate_dt / bwd_plane / bwd_coefs), `saved` contract: cnsn_layout.h —
// exactly what the cluster-resident kernels evaluate, so forward and backward strategies stay interchangeable.
#pragma once
#include "cnsn_mono_kernels.h"

namespace cnsn {

struct MonoCnArgs {
    MonoArgs m;
    int Wd;      // plane width (element index -> row, column)
    Box cb, sb;  // content / style box (whole plane when the call has none)
};

// per-plane LDS arrays (cap entries each)
enum MonoCnFwdRow { MF_MU_C = 0, MF_M2C, MF_MU_O, MF_M2O, MF_MU_S, MF_M2S, MF_A_IN, MF_XR, MF_B_IN, MF_A_OUT, MF_B_OUT, MF_ROWS };
enum MonoCnBwdRow {
    MB_SI = 0, MB_SO, MB_FA_IN, MB_FXR, MB_FB_IN, MB_FA_OUT, MB_FB_OUT,  // shifts of the sums, forward coefficients (ReLU mask)
    MB_S1I, MB_S2I, MB_S1O, MB_S2O,                                      // the four sums
    MB_EMU, MB_ESIG,                                                     // what a plane sends to its style source
    MB_C0,                                                               // 11 coefficients of dx follow
    MB_ROWS = MB_C0 + 11
};

__host__ __device__ inline size_t mono_cn_lds_bytes(int cap, bool backward) {
    return (size_t)cap * 4 * (backward ? (MB_ROWS + 1) : MF_ROWS) + (size_t)kMonoWaves * 4 * 8 + 16 * 8;
}

// bit q of the result: element vl*VEC + q of the plane lies in the box
template <int VEC>
__device__ __forceinline__ unsigned mono_box_bits(const Box& b, int vl, int nvec, int Wd) {
    unsigned m = 0;
    if (vl < nvec) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int e = vl * VEC + q, r = e / Wd, c = e - r * Wd;
            m |= (b.has(r, c) ? 1u : 0u) << q;
        }
    }
    return m;
}

// ================================================================================================
// forward
// ================================================================================================
template <typename T, int VEC, int LPP, int RMAX, bool EPI>
__global__ __launch_bounds__(kMonoBlock) void mono_cn_fwd_kernel(MonoCnArgs ca, const T* __restrict__ x,
                                                                     const T* __restrict__ addend, T* __restrict__ y,
                                                                     const int64_t* __restrict__ perm, GateDev gg,
                                                                     double* __restrict__ saved, int add, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MonoArgs& ma = ca.m;
    MidArgs a = ma.mid;
    a.sn_two = 0;
    const int N = a.N, C = a.C;
    const int cap = kMonoWaves * ma.R * (64 / LPP);
    const MonoWalk wk(C);
    if (wk.j < wk.count) {
        const int c = wk.start + wk.j;
        float* st = (float*)smem;  // [MF_ROWS][cap]
        double* red = (double*)(st + (size_t)MF_ROWS * cap);
        double* par = red + kMonoWaves * 4;
        const size_t P = (size_t)N * C;
        MonoGeom<T, VEC, LPP> g(ma);
        g.set_channel(c);
        const unsigned cm = mono_box_bits<VEC>(ca.cb, g.vl, ma.nvec, ca.Wd);
        const unsigned sm = mono_box_bits<VEC>(ca.sb, g.vl, ma.nvec, ca.Wd);

        const int n = threadIdx.x;
        const bool act = n < N;
        const int q_src = act ? (int)perm[n] : 0;  // style source of plane n (same channel, cnsn.py:62,66)
        if (threadIdx.x == 0 && a.sn_active) {
            par[0] = gg.w[2 * c];
            par[1] = gg.w[2 * c + 1];
            par[2] = gg.gamma[c];
            par[3] = gg.beta[c];
            par[4] = gg.run_mean[c];
            par[5] = gg.run_var[c];
        }

        // ---- the only read of x (+ addend)
        MRaw<T, VEC> d[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) d[r] = mload<T, VEC>(g.rsrc(x, c, r), g.off(r));
        if constexpr (EPI) {
            if (add == ADD_PRE) {
                constexpr int CH = (RMAX * VEC * (int)sizeof(T) > 128) ? RMAX / 2 : RMAX;
#pragma unroll
                for (int r0 = 0; r0 < RMAX; r0 += CH) {
                    MRaw<T, VEC> q[CH];
#pragma unroll
                    for (int r = 0; r < CH; ++r) q[r] = mload<T, VEC>(g.rsrc(addend, c, r0 + r), g.off(r0 + r));
#pragma unroll
                    for (int r = 0; r < CH; ++r) d[r0 + r] = madd<T, VEC>(d[r0 + r], q[r]);
                }
            }
        }

        // ---- exact two-pass statistics of the three regions of every plane
        const float Mc = (float)a.Mc, Ms = (float)a.Ms;
        const int Mo_i = a.M - a.Mc;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            float sc = 0.f, so = 0.f, ss = 0.f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float f = melem<T, VEC>(d[r], q);  // lanes that are not ok() loaded zeros and carry no mask bit
                const bool ic = (cm >> q) & 1u, is = (sm >> q) & 1u;
                sc += ic ? f : 0.f;
                so += ic ? 0.f : f;
                ss += is ? f : 0.f;
            }
            const float mc = mono_group_sum<LPP>(sc) / Mc;
            const float mo = Mo_i > 0 ? mono_group_sum<LPP>(so) / (float)Mo_i : 0.f;
            const float ms = mono_group_sum<LPP>(ss) / Ms;
            mono_forget(d[r]);
            float qc = 0.f, qo = 0.f, qs = 0.f;
            if (g.ok(r)) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float f = melem<T, VEC>(d[r], q);
                    const bool ic = (cm >> q) & 1u, is = (sm >> q) & 1u;
                    const float tc = f - mc, to = f - mo, ts = f - ms;
                    qc += ic ? tc * tc : 0.f;
                    qo += ic ? 0.f : to * to;
                    qs += is ? ts * ts : 0.f;
                }
            }
            qc = mono_group_sum<LPP>(qc);
            qo = mono_group_sum<LPP>(qo);
            qs = mono_group_sum<LPP>(qs);
            if (g.vl == 0 && g.ok(r)) {
                const int pn = g.plane(r);
                st[MF_MU_C * cap + pn] = mc;
                st[MF_M2C * cap + pn] = qc;
                st[MF_MU_O * cap + pn] = mo;
                st[MF_M2O * cap + pn] = qo;
                st[MF_MU_S * cap + pn] = ms;
                st[MF_M2S * cap + pn] = qs;
            }
        }
        __syncthreads();

        // ---- per-plane algebra, a thread per plane: CrossNorm against the style source, then SelfNorm's gate
        using Rr = float;
        FwdPlaneT<Rr> f{};
        double zg = 0.0;
        if (act) {
            MomentsT<Rr> o;
            o.mu_c = st[MF_MU_C * cap + n];
            o.M2c = st[MF_M2C * cap + n];
            o.mu_o = st[MF_MU_O * cap + n];
            o.M2o = st[MF_M2O * cap + n];
            o.mu_s = st[MF_MU_S * cap + n];
            o.M2s = st[MF_M2S * cap + n];
            f = fwd_plane<Rr>(a, o, st[MF_MU_S * cap + q_src], st[MF_M2S * cap + q_src]);
            if (a.sn_active) zg = par[0] * (double)f.mu_p + par[1] * (double)f.sig_p;
        }
        double mg = 0.0, rg = 1.0;
        if (a.sn_active) {
            mg = par[4];
            if (a.sn_training) {
                double sz[2] = {act ? zg : 0.0, act ? zg * zg : 0.0};
                mono_block_sum<2>(sz, red);
                mg = sz[0] * a.inv_n;
                double vg = sz[1] * a.inv_n - mg * mg;
                vg = vg > 0.0 ? vg : 0.0;
                rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
                if (threadIdx.x == 0) {
                    const double mom_ = a.momentum, unb = a.unbias_n;
                    gg.run_mean[c] = (float)((1.0 - mom_) * par[4] + mom_ * mg);
                    gg.run_var[c] = (float)((1.0 - mom_) * par[5] + mom_ * vg * unb);
                    if (c == 0) bump_batches_tracked(gg.nbt);
                }
            } else {
                rg = (double)__builtin_amdgcn_rsqf((float)par[5] + a.eps_bn);
            }
            if (saved && threadIdx.x == 0) {
                saved[SV_ROWS * P + c] = rg;
                saved[SV_ROWS * P + C + c] = 1.0;
            }
        }
        if (act) {
            double zhg = 0.0;
            Rr gt = 1.f;
            if (a.sn_active) {
                zhg = (zg - mg) * rg;
                gt = sigmoid_r<Rr>((Rr)(par[2] * zhg + par[3]));
            }
            const FwdCoefs cf = fwd_coefs<Rr>(a, f, gt, Rr(1));
            if (saved) {
                const SvRec p = sv_rec(n, c, N);
                store_fwd_plane<Rr>(saved, P, p, f, 1);
                saved[sv_at(p, SV_G)] = gt;
                saved[sv_at(p, SV_ZH_G)] = zhg;
                saved[sv_at(p, SV_F)] = 1.0;
                saved[sv_at(p, SV_ZH_F)] = 0.0;
                if (a.save_coefs) store_fwd_coefs(saved, p, cf);
            }
            st[MF_A_IN * cap + n] = cf.a_in;
            st[MF_XR * cap + n] = cf.xr;
            st[MF_B_IN * cap + n] = cf.b_in;
            st[MF_A_OUT * cap + n] = cf.a_out;
            st[MF_B_OUT * cap + n] = cf.b_out;
        }
        __syncthreads();

        // ---- apply from registers, the only write of y
        g.refresh();
#pragma unroll
        for (int r = 0; r < RMAX; ++r) mono_forget(d[r]);
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int pn = g.plane(r);
            const float a_in = st[MF_A_IN * cap + pn], xr = st[MF_XR * cap + pn], b_in = st[MF_B_IN * cap + pn],
                        a_out = st[MF_A_OUT * cap + pn], b_out = st[MF_B_OUT * cap + pn];
            float ov[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float f = melem<T, VEC>(d[r], q);
                ov[q] = ((cm >> q) & 1u) ? fmaf(a_in, f - xr, b_in) : fmaf(a_out, f, b_out);
                if constexpr (EPI) ov[q] = relu ? fmaxf(ov[q], 0.f) : ov[q];
            }
            mstore<T, VEC>(g.rsrc(y, c, r), g.off(r), mpack<T, VEC>(ov));
        }
    }
}

// ================================================================================================
// backward
// ================================================================================================
template <typename T, int VEC, int LPP, int RMAX, bool EPI>
__global__ __launch_bounds__(kMonoBlock) void mono_cn_bwd_kernel(MonoCnArgs ca, const T* __restrict__ gy, const T* __restrict__ x,
                                                                     const T* __restrict__ addend, T* __restrict__ dx,
                                                                     const int64_t* __restrict__ perm, GateDev gg,
                                                                     GateGradDev dgr, const double* __restrict__ saved,
                                                                     int add, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MonoArgs& ma = ca.m;
    MidArgs a = ma.mid;
    a.sn_two = 0;
    const int N = a.N, C = a.C;
    const int cap = kMonoWaves * ma.R * (64 / LPP);
    const MonoWalk wk(C);
    if (wk.j < wk.count) {
        const int c = wk.start + wk.j;
        float* st = (float*)smem;                          // [MB_ROWS][cap]
        int* iperm = (int*)(st + (size_t)MB_ROWS * cap);   // [cap] inverse permutation
        double* red = (double*)(iperm + cap);
        const size_t P = (size_t)N * C;
        MonoGeom<T, VEC, LPP> g(ma);
        g.set_channel(c);
        const unsigned cm = mono_box_bits<VEC>(ca.cb, g.vl, ma.nvec, ca.Wd);
        const unsigned sm = mono_box_bits<VEC>(ca.sb, g.vl, ma.nvec, ca.Wd);

        // ---- what the sums need from `saved`, a thread per plane, ahead of the bulk loads
        const int n = threadIdx.x;
        const bool act = n < N;
        const SvRec pme = sv_rec(act ? n : 0, c, N);
        if (act) {
            iperm[(int)perm[n]] = n;  // plane perm[n] lent its statistics to plane n
            st[MB_SI * cap + n] = (float)saved[sv_at(pme, SV_MU_C)];
            st[MB_SO * cap + n] = (float)saved[sv_at(pme, SV_MU_O)];
            if (EPI && relu) {
#pragma unroll
                for (int k = 0; k < FC_ROWS; ++k) st[(MB_FA_IN + k) * cap + n] = (float)saved[sv_at(pme, SV_FC0 + k)];
            }
        }
        float w_g0 = 0.f, w_g1 = 0.f, gam_g = 0.f;
        double rs_g = 1.0;
        if (a.sn_active) {
            w_g0 = gg.w[2 * c];
            w_g1 = gg.w[2 * c + 1];
            gam_g = gg.gamma[c];
            rs_g = saved[SV_ROWS * P + c];
        }

        // ---- the only reads of G and x (+ addend)
        MRaw<T, VEC> dg_[RMAX], dx_[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            dg_[r] = mload<T, VEC>(g.rsrc(gy, c, r), g.off(r));
            dx_[r] = mload<T, VEC>(g.rsrc(x, c, r), g.off(r));
        }
        if constexpr (EPI) {
            if (add == ADD_PRE) {
                constexpr int CH = (RMAX * VEC * (int)sizeof(T) > 64) ? RMAX / 2 : RMAX;
#pragma unroll
                for (int r0 = 0; r0 < RMAX; r0 += CH) {
                    MRaw<T, VEC> q[CH];
#pragma unroll
                    for (int r = 0; r < CH; ++r) q[r] = mload<T, VEC>(g.rsrc(addend, c, r0 + r), g.off(r0 + r));
#pragma unroll
                    for (int r = 0; r < CH; ++r) dx_[r0 + r] = madd<T, VEC>(dx_[r0 + r], q[r]);
                }
            }
        }
        __syncthreads();  // the staged rows and the inverse permutation are in LDS
        g.refresh();

        // ---- ReLU mask and the four per-plane sums (shifted by the saved means, as pass A' does)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int pn = g.plane(r);
            const float si = st[MB_SI * cap + pn], so = st[MB_SO * cap + pn];
            if constexpr (EPI) {
                if (relu) {
                    const float a_in = st[MB_FA_IN * cap + pn], xr = st[MB_FXR * cap + pn], b_in = st[MB_FB_IN * cap + pn],
                                a_out = st[MB_FA_OUT * cap + pn], b_out = st[MB_FB_OUT * cap + pn];
                    float gm[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float X = melem<T, VEC>(dx_[r], q);
                        const float t = ((cm >> q) & 1u) ? fmaf(a_in, X - xr, b_in) : fmaf(a_out, X, b_out);
                        gm[q] = relu_open_r<T>(t) ? melem<T, VEC>(dg_[r], q) : 0.f;
                    }
                    dg_[r] = mpack<T, VEC>(gm);
                }
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (g.ok(r)) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float G = melem<T, VEC>(dg_[r], q), X = melem<T, VEC>(dx_[r], q);
                    const bool ic = (cm >> q) & 1u;
                    acc[0] += ic ? G : 0.f;
                    acc[1] += ic ? G * (X - si) : 0.f;
                    acc[2] += ic ? 0.f : G;
                    acc[3] += ic ? 0.f : G * (X - so);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = mono_group_sum<LPP>(acc[k]);
            if (g.vl == 0 && g.ok(r)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) st[(MB_S1I + k) * cap + pn] = acc[k];
            }
        }
        __syncthreads();

        // ---- gate / BatchNorm backward and the statistic gradients, a thread per plane
        using Rr = float;
        double r_mu = 0, r_zhg = 0;
        Rr r_mup = 0, r_sigp = 1, r_g = 1;
        CnRowsT<Rr> cr{};
        BwdSumsT<Rr> sums{};
        Rr dtg = 0.f, dtf = 0.f;
        if (act) {
            r_mu = saved[sv_at(pme, SV_MU_C)];
            r_mup = (Rr)saved[sv_at(pme, SV_MU_P)];
            r_sigp = (Rr)saved[sv_at(pme, SV_SIG_P)];
            r_g = (Rr)saved[sv_at(pme, SV_G)];
            r_zhg = saved[sv_at(pme, SV_ZH_G)];
            cr = load_cn_rows<Rr>(a, saved, pme, r_mu);
            sums = fix_sums<Rr>(a, st[MB_S1I * cap + n], st[MB_S2I * cap + n], st[MB_S1O * cap + n], st[MB_S2O * cap + n], r_mu,
                                (double)cr.mu_o);
            if (a.sn_active) gate_dt<Rr>(a, sums, cr.a1, cr.m_in, cr.mu_o, r_mup, r_g, Rr(1), dtg, dtf);
        }
        BnBwd b{};
        double s4[2] = {(double)dtg, (double)dtg * r_zhg};
        if (a.sn_active) {
            mono_block_sum<2>(s4, red);
            b.s_dt_g = s4[0];
            b.s_dtz_g = s4[1];
            b.wg0 = w_g0;
            b.wg1 = w_g1;
            b.kg = (double)gam_g * rs_g;
        }
        BwdPlaneT<Rr> o{};
        double sw[2] = {0, 0};
        if (act) {
            o = bwd_plane<Rr>(a, b, sums, (double)dtg, 0.0, r_zhg, 0.0, r_g, Rr(1), cr.aa, cr.a1, cr.m_in, r_mup, r_sigp, cr.sig_c,
                              cr.M2c);
            sw[0] = (double)o.dz_g * (double)r_mup;
            sw[1] = (double)o.dz_g * (double)r_sigp;
            st[MB_EMU * cap + n] = o.Emu;
            st[MB_ESIG * cap + n] = o.Esig;
        }
        if (a.sn_active) {
            mono_block_sum<2>(sw, red);  // (also the barrier that makes Emu / Esig of every plane visible)
            if (threadIdx.x == 0) {
                dgr.dgamma[c] = (float)s4[1];
                dgr.dbeta[c] = (float)s4[0];
                dgr.dw[2 * c] = (float)sw[0];
                dgr.dw[2 * c + 1] = (float)sw[1];
            }
        } else {
            __syncthreads();
        }
        if (act) {
            const int src = iperm[n];  // the plane that used (n, c) as its style
            const BwdCoefs k = bwd_coefs<Rr>(a, o, st[MB_EMU * cap + src], st[MB_ESIG * cap + src], r_g, cr.a1, cr.m_in, r_mup, r_mu,
                                             cr.sig_c, cr.mu_s, cr.sig_s);
            float* pc = st + (size_t)MB_C0 * cap + n;
            pc[0 * cap] = k.cG_in;
            pc[1 * cap] = k.cX_in;
            pc[2 * cap] = k.xr_in;
            pc[3 * cap] = k.c0_in;
            pc[4 * cap] = k.cG_out;
            pc[5 * cap] = k.cX_out;
            pc[6 * cap] = k.xr_out;
            pc[7 * cap] = k.c0_out;
            pc[8 * cap] = k.eS;
            pc[9 * cap] = k.xs;
            pc[10 * cap] = k.e0;
        }
        __syncthreads();

        // ---- dx from registers, the only write
        const bool boxed = a.boxed != 0;
        g.refresh();
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            mono_forget(dg_[r]);
            mono_forget(dx_[r]);
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float* pc = st + (size_t)MB_C0 * cap + g.plane(r);
            const float cG_i = pc[0 * cap], cX_i = pc[1 * cap], xr_i = pc[2 * cap], c0_i = pc[3 * cap];
            const float cG_o = pc[4 * cap], cX_o = pc[5 * cap], xr_o = pc[6 * cap], c0_o = pc[7 * cap];
            const float eS = pc[8 * cap], xs = pc[9 * cap], e0 = pc[10 * cap];
            float ov[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float G = melem<T, VEC>(dg_[r], q), X = melem<T, VEC>(dx_[r], q);
                float v = ((cm >> q) & 1u) ? fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i)) : fmaf(cG_o, G, fmaf(cX_o, X - xr_o, c0_o));
                // (without boxes bwd_coefs has already folded the style term into the in-region coefficients)
                v += (boxed && ((sm >> q) & 1u)) ? fmaf(eS, X - xs, e0) : 0.f;
                ov[q] = v;
            }
            mstore<T, VEC>(g.rsrc(dx, c, r), g.off(r), mpack<T, VEC>(ov));
            if ((r & (MONO_ILP - 1)) == MONO_ILP - 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace cnsn
