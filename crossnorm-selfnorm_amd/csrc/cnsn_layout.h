// Layout of the per-plane side arrays shared by the two strategies (`saved` written by either
// forward and read by either backward; coefficient rows consumed by the apply kernels).
#pragma once
#include "cnsn_algebra.h"

namespace cnsn {

// `saved`: SV_ROWS doubles per plane, stored CHANNEL BY CHANNEL and, within a channel, ROW BY ROW over the batch: value
// `row` of plane (n, c) is element (c*SV_ROWS + row)*N + n; two rows of C (BatchNorm rstd of g and f) follow at
// SV_ROWS*P.  Why: every consumer that needs more than its own plane walks ONE channel over the batch (the BatchNorm1d
// over N of cnsn.py:121,138; the style permutation of cnsn.py:62) with one thread per instance — in this layout every
// such load or store is 64 consecutive doubles.  In tensor order with one record per plane (round 1) the same walk
// touched one 128-byte line per 8-byte field: at the north-star shape four times as many memory requests as the planes
// themselves (round 2: the backward of the resident kernels lost 40 us to it).  SvRec keeps record positions and plane
// numbers (p = n*C + c, the order of every other side array) apart at compile time.
enum SavedRow {
    // ---- rows every mode needs, first and contiguous: a SelfNorm-only forward writes just these seven (one or
    //      two 64-byte sectors per plane instead of scattered 8-byte writes over a 160-byte record — for planes of
    //      a few hundred bytes the side-array traffic is otherwise comparable to the tensor's own)
    SV_MU_C = 0,   // mean inside the content box (whole plane without one)
    SV_MU_P,       // post-CrossNorm whole-plane mean  (SelfNorm's input statistic; = SV_MU_C without CrossNorm)
    SV_SIG_P,      // post-CrossNorm whole-plane std, eps_sn
    SV_G,          // gate g
    SV_ZH_G,       // normalised pre-activation of g
    SV_F,          // gate f (two-gate form; 1 otherwise)
    SV_ZH_F,
    // ---- CrossNorm only: NOT WRITTEN when the call has no CrossNorm — readers substitute the constants of
    //      load_cn_rows() (a = a1 = 1, m_in = mu_s = mu_c, mu_o = 0, ...)
    SV_MU_O,       // mean outside the content box
    SV_M2C,        // sum of squared deviations inside the content box
    SV_SIG_C,      // sqrt(var_c + eps_cn)
    SV_MU_S,       // this plane's own style-box mean   (what it lends as a style source)
    SV_SIG_S,      // this plane's own style-box std
    SV_A,          // sig_s[q] / sig_c
    SV_A1,         // lam + (1-lam)*a : slope applied inside the content box
    SV_M_IN,       // mean of the CrossNorm output inside the content box
    // the forward's apply coefficients (floats, exact in a double): written only for a ReLU-fused call, whose
    // backward re-evaluates the forward affine bit-for-bit to recover the ReLU mask without reading y
    SV_FC0,
    SV_ROWS = SV_FC0 + 5
};

struct SvRec {
    size_t base;  // position of row 0 of the plane: c*SV_ROWS*N + n
    int N;        // distance between two rows of the plane
};
__host__ __device__ inline SvRec sv_rec(int n, int c, int N) { return SvRec{(size_t)c * SV_ROWS * (size_t)N + (size_t)n, N}; }
// from a plane number in tensor order (p = n*C + c)
__host__ __device__ inline SvRec sv_rec_of_plane(size_t p, int N, int C) {
    const size_t n = p / (size_t)C;
    return sv_rec((int)n, (int)(p - n * (size_t)C), N);
}
__host__ __device__ inline size_t sv_at(SvRec r, int row) { return r.base + (size_t)row * (size_t)r.N; }

// rows of the forward coefficient block handed to apply_fwd_kernel
enum FwdCoefRow { FC_A_IN = 0, FC_XR, FC_B_IN, FC_A_OUT, FC_B_OUT, FC_ROWS };

// rows of the backward scratch written by mid_bwd_a and read by mid_bwd_b
enum BwdTmpRow { BT_DT_G = 0, BT_DT_F, BT_DMU_P, BT_K, BT_DMU_C, BT_DSIG_C, BT_E_MU, BT_E_SIG, BT_ROWS };

// rows of the backward coefficient block handed to apply_bwd_kernel
enum BwdCoefRow {
    BC_CG_IN = 0, BC_CX_IN, BC_XR_IN, BC_C0_IN, BC_CG_OUT, BC_CX_OUT, BC_XR_OUT, BC_C0_OUT, BC_ES, BC_XS, BC_E0,
    BC_ROWS
};

template <typename R>
__device__ __forceinline__ void store_fwd_plane(double* __restrict__ saved, size_t P, SvRec p,
                                                const FwdPlaneT<R>& f, int cn_active) {
    saved[sv_at(p, SV_MU_C)] = f.mu_c;
    saved[sv_at(p, SV_MU_P)] = f.mu_p;
    saved[sv_at(p, SV_SIG_P)] = f.sig_p;
    if (cn_active) {
        saved[sv_at(p, SV_MU_O)] = f.mu_o;
        saved[sv_at(p, SV_M2C)] = f.M2c;
        saved[sv_at(p, SV_SIG_C)] = f.sig_c;
        saved[sv_at(p, SV_MU_S)] = f.mu_s;
        saved[sv_at(p, SV_SIG_S)] = f.sig_s;
        saved[sv_at(p, SV_A)] = f.aa;
        saved[sv_at(p, SV_A1)] = f.a1;
        saved[sv_at(p, SV_M_IN)] = f.m_in;
    }
}

// the CrossNorm-only rows of plane p, or what they amount to when the call has no CrossNorm
template <typename R>
struct CnRowsT {
    R mu_o, M2c, sig_c, sig_s, aa, a1, m_in;
    double mu_s;
};
template <typename R>
__device__ __forceinline__ CnRowsT<R> load_cn_rows(const MidArgs& a, const double* __restrict__ saved, SvRec p,
                                                   double mu_c) {
    CnRowsT<R> r;
    if (a.cn_active) {
        r.mu_o = (R)saved[sv_at(p, SV_MU_O)];
        r.M2c = (R)saved[sv_at(p, SV_M2C)];
        r.sig_c = (R)saved[sv_at(p, SV_SIG_C)];
        r.mu_s = saved[sv_at(p, SV_MU_S)];
        r.sig_s = (R)saved[sv_at(p, SV_SIG_S)];
        r.aa = (R)saved[sv_at(p, SV_A)];
        r.a1 = (R)saved[sv_at(p, SV_A1)];
        r.m_in = (R)saved[sv_at(p, SV_M_IN)];
    } else {
        r.mu_o = R(0);
        r.M2c = R(0);
        r.sig_c = R(1);
        r.mu_s = mu_c;
        r.sig_s = R(1);
        r.aa = R(1);
        r.a1 = R(1);
        r.m_in = (R)mu_c;
    }
    return r;
}

__device__ __forceinline__ void store_fwd_coefs(double* __restrict__ saved, SvRec p, const FwdCoefs& k) {
    saved[sv_at(p, SV_FC0 + FC_A_IN)] = k.a_in;
    saved[sv_at(p, SV_FC0 + FC_XR)] = k.xr;
    saved[sv_at(p, SV_FC0 + FC_B_IN)] = k.b_in;
    saved[sv_at(p, SV_FC0 + FC_A_OUT)] = k.a_out;
    saved[sv_at(p, SV_FC0 + FC_B_OUT)] = k.b_out;
}

}  // namespace cnsn
