// Channel-local strategy for SMALL planes without CrossNorm (SelfNorm alone — every site of the ResNet-50
// configs — optionally with the residual block's add and ReLU): ONE launch per direction, every byte touched once,
// and no inter-workgroup exchange at all.
//
// When a plane is small (7x7, 8x8, 14x14 ...) all N planes of a channel — even of a few adjacent channels — fit in
// ONE workgroup's LDS (N*M*b bytes per channel: 25 KiB for (256,.,7,7) bf16).  SelfNorm's BatchNorm1d couples only
// the N planes of a channel, so a workgroup that owns a whole channel group needs nobody else:
//     forward : pieces (n, c0..c0+CG-1) of x [+ addend] -> LDS; per-plane exact two-pass statistics (16 lanes per
//               plane); BatchNorm over n / gate per plane (a thread per plane, cnsn_algebra.h in float, batch sums
//               in double); apply from LDS -> y.
//     backward: G, x [+ addend] -> LDS; ReLU mask + per-plane sums; gate / BatchNorm backward; dx.
// Compared with the two-pass kernels this removes the three mid kernels and every per-plane side array except
// `saved` (for 98-byte planes those arrays are as large as the tensor itself), and compared with the cluster-
// resident kernels it removes the exchange whose latency dominates when a plane is only a few hundred bytes.
//
// Memory access: for fixed n the CG channels are one contiguous piece of CG*M*b bytes; pieces are copied with the
// widest vector (16/8/4/2 bytes) that divides both the piece and the distance between pieces, consecutive lanes on
// consecutive vectors.  The LDS image is plane-major: plane (n, cc) starts at element (n*CG + cc)*M.
#pragma once
#include "../../include/cnsn_hip.h"
#include "cnsn_algebra.h"
#include "cnsn_device.h"
#include "cnsn_fused_stream_kernels.h"
#include "cnsn_layout.h"

namespace cnsn {

struct LocalArgs {
    MidArgs mid;
    int CG;          // channels per workgroup
    int W;           // bytes per copy vector
    int piece_vecs;  // vectors per piece = CG*M*b / W
    int planes;      // N * CG
    float inv_m;     // 1/M (element index -> plane index)
};

// tuning builds (-DCNSN_LPROF): per-phase wall-clock stamps of the first and last workgroup, printed by the kernel
#ifdef CNSN_LPROF
#define LSTAMP() t_[ti_++] = (long long)wall_clock64()
#else
#define LSTAMP() \
    do {         \
    } while (0)
#endif

constexpr int kLocalMaxCG = 8;
// threads per workgroup (template parameter LB): the phases of a workgroup run one after the other and each is
// bound by latency.  A small image (many workgroups per CU) gets 256 threads and lets the workgroups of a CU overlap;
// a large image (one or two workgroups per CU) gets 1024 threads so that the parallelism is inside the workgroup.
constexpr int kLocalBigBlock = 1024;

__host__ __device__ inline size_t local_align(size_t v) { return (v + 15) & ~(size_t)15; }

// copy vector v (of `W` bytes) of the channel group between HBM and registers
template <int W>
struct RawW;
template <>
struct RawW<16> {
    typedef unsigned type __attribute__((ext_vector_type(4)));
};
template <>
struct RawW<8> {
    typedef unsigned type __attribute__((ext_vector_type(2)));
};
template <>
struct RawW<4> {
    typedef unsigned type;
};
template <>
struct RawW<2> {
    typedef unsigned short type;
};

// x [+ addend] -> LDS, vector by vector (U loads in flight per thread)
template <typename T, int W, bool ADD, int LB>
__device__ __forceinline__ void local_stage(const LocalArgs& la, const T* __restrict__ src, const T* __restrict__ add,
                                            int c0, char* lds) {
    using V = typename RawW<W>::type;
    // loads in flight per thread: up to 256 bytes (a workgroup with a large image is alone on its CU, so the memory-level
    // parallelism has to come from within it)
    constexpr int U = LB == 256 ? (W == 16 ? 16 : 32) : (W == 16 ? 4 : 8), E = W / (int)sizeof(T);
    const MidArgs& a = la.mid;
    const int total = a.N * la.piece_vecs;
    const size_t row = (size_t)a.C * a.M * sizeof(T);  // bytes between the pieces of consecutive instances
    const char* s = (const char*)src + (size_t)c0 * a.M * sizeof(T);
    const char* d = ADD ? (const char*)add + (size_t)c0 * a.M * sizeof(T) : nullptr;
    for (int v0 = threadIdx.x; v0 < total; v0 += LB * U) {
        V r[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * LB;
            if (v < total) {
                const int n = v / la.piece_vecs, k = v - n * la.piece_vecs;
                const size_t off = (size_t)n * row + (size_t)k * W;
                r[u] = __builtin_nontemporal_load((const V*)(s + off));
                if constexpr (ADD) q[u] = __builtin_nontemporal_load((const V*)(d + off));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * LB;
            if (v < total) {
                if constexpr (ADD) {  // the op's input is x + addend, a value of the tensor's own type
                    T xe[E], ae[E];
                    __builtin_memcpy(xe, &r[u], W);
                    __builtin_memcpy(ae, &q[u], W);
#pragma unroll
                    for (int e = 0; e < E; ++e) xe[e] = from_float<T>(to_float(xe[e]) + to_float(ae[e]));
                    __builtin_memcpy(&r[u], xe, W);
                }
                *(V*)(lds + (size_t)v * W) = r[u];
            }
        }
    }
}

// apply a per-plane affine map to the staged image and write it out: emit(plane, elems in, elems out)
template <typename T, int W, int LB, typename F>
__device__ __forceinline__ void local_emit(const LocalArgs& la, T* __restrict__ dst, int c0, F&& f) {
    using V = typename RawW<W>::type;
    constexpr int E = W / (int)sizeof(T);
    const MidArgs& a = la.mid;
    const int total = a.N * la.piece_vecs;
    const size_t row = (size_t)a.C * a.M * sizeof(T);
    char* d = (char*)dst + (size_t)c0 * a.M * sizeof(T);
    for (int v = threadIdx.x; v < total; v += LB) {
        const int n = v / la.piece_vecs, k = v - n * la.piece_vecs;
        const int e0 = v * E;  // first element of the vector in the plane-major image
        T o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = e0 + e;
            int pl = (int)(((float)idx + 0.5f) * la.inv_m);  // idx / M (exact for idx < 2^22)
            o[e] = f(pl, idx);
        }
        V r;
        __builtin_memcpy(&r, o, W);
        __builtin_nontemporal_store(r, (V*)(d + (size_t)n * row + (size_t)k * W));
    }
}

// sum of NACC doubles over the 256-thread GROUP a thread belongs to (LB/256 groups per workgroup, each working on
// its own channel); every thread of the workgroup must call it (workgroup barriers).  red: [LB/64][NACC] doubles.
template <int NACC>
__device__ __forceinline__ void local_group_sum(double (&acc)[NACC], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g0 = wave & ~3;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[k] = v;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) red[wave * NACC + k] = acc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NACC; ++k)
        acc[k] = (red[g0 * NACC + k] + red[(g0 + 1) * NACC + k]) + (red[(g0 + 2) * NACC + k] + red[(g0 + 3) * NACC + k]);
}

// ================================================================================================
// forward
// ================================================================================================
template <typename T, int W, bool EPI, int LB>
__global__ __launch_bounds__(LB) void local_fwd_kernel(LocalArgs la, const T* __restrict__ x,
                                                           const T* __restrict__ addend, T* __restrict__ y, GateDev gg,
                                                           GateDev gf, double* __restrict__ saved, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = la.mid;
    const int N = a.N, C = a.C, M = a.M, CG = la.CG, planes = la.planes;
    const int c0 = blockIdx.x * CG;
    T* img = (T*)smem;
    float* pmu = (float*)(smem + local_align((size_t)planes * M * sizeof(T)));  // [planes]: mean, later slope
    float* pm2 = pmu + planes;                                                   // [planes]: M2, later offset
    double* par = (double*)((char*)pmu + local_align((size_t)2 * planes * 4));    // [CG][16] per-channel parameters
    double* red = par + kLocalMaxCG * 16;
    const size_t P = (size_t)N * C;
#ifdef CNSN_LPROF
    long long t_[8];
    int ti_ = 0;
#endif
    LSTAMP();

    // per-channel parameters -> LDS (issued before the bulk loads)
    if ((int)threadIdx.x < CG) {
        const int c = c0 + threadIdx.x;
        double* q = par + threadIdx.x * 16;
        q[0] = gg.w[2 * c];
        q[1] = gg.w[2 * c + 1];
        q[2] = gg.gamma[c];
        q[3] = gg.beta[c];
        q[4] = gg.run_mean[c];
        q[5] = gg.run_var[c];
        if (a.sn_two) {
            q[6] = gf.w[2 * c];
            q[7] = gf.w[2 * c + 1];
            q[8] = gf.gamma[c];
            q[9] = gf.beta[c];
            q[10] = gf.run_mean[c];
            q[11] = gf.run_var[c];
        }
    }
    if (EPI && addend)
        local_stage<T, W, true, LB>(la, x, addend, c0, smem);
    else
        local_stage<T, W, false, LB>(la, x, nullptr, c0, smem);
    __syncthreads();
    LSTAMP();

    // ---- exact two-pass statistics of every plane: 16 lanes per plane, four planes in flight per 16-lane group
    //      (the loop is bound by LDS latency, not bandwidth); planes of up to 64 elements are read once
    {
        const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
        constexpr int PU = 4, G16 = LB / 16;
        for (int pl0 = grp; pl0 < planes; pl0 += G16 * PU) {
            if (M <= 64) {
                float v[PU][4];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int pl = pl0 + u * G16;
                    const T* pp = img + (size_t)(pl < planes ? pl : pl0) * M;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int e = l16 + 16 * k;
                        v[u][k] = e < M ? to_float(pp[e]) : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int pl = pl0 + u * G16;
                    const float mean = row16_sum((v[u][0] + v[u][1]) + (v[u][2] + v[u][3])) / (float)M;
                    float q = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = (l16 + 16 * k < M) ? v[u][k] - mean : 0.f;
                        q = fmaf(t, t, q);
                    }
                    q = row16_sum(q);
                    if (l16 == 0 && pl < planes) {
                        pmu[pl] = mean;
                        pm2[pl] = q;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int pl = pl0 + u * G16;
                    if (pl >= planes) break;  // uniform within the 16-lane group
                    const T* pp = img + (size_t)pl * M;
                    float s = 0.f;
                    for (int e = l16; e < M; e += 16) s += to_float(pp[e]);
                    const float mean = row16_sum(s) / (float)M;
                    float q = 0.f;
                    for (int e = l16; e < M; e += 16) {
                        const float t = to_float(pp[e]) - mean;
                        q = fmaf(t, t, q);
                    }
                    q = row16_sum(q);
                    if (l16 == 0) {
                        pmu[pl] = mean;
                        pm2[pl] = q;
                    }
                }
            }
        }
    }
    __syncthreads();
    LSTAMP();

    // ---- gates: BatchNorm1d over the N planes of each channel; each 256-thread group of the workgroup takes
    //      one channel at a time (idle groups still walk through the barriers)
    using R = float;
    constexpr int SUB = LB / 256;
    const int gidx = threadIdx.x >> 8, gt = threadIdx.x & 255;
    for (int cc0 = 0; cc0 < CG; cc0 += SUB) {
        const bool act = cc0 + gidx < CG;
        const int cc = act ? cc0 + gidx : 0;
        const int c = c0 + cc;
        const double* q = par + cc * 16;
        const double wg0 = q[0], wg1 = q[1], wf0 = q[6], wf1 = q[7];
        auto plane_of = [&](int n) {
            MomentsT<R> o;
            const int pl = n * CG + cc;
            o.mu_c = o.mu_s = pmu[pl];
            o.M2c = o.M2s = pm2[pl];
            o.mu_o = o.M2o = 0.f;
            return fwd_plane<R>(a, o, 0.f, 0.f);
        };
        double mg = q[4], mf = q[10], rg, rf;
        if (a.sn_training) {
            const FwdPlaneT<R> f0 = plane_of(0);
            const double zs_g = wg0 * (double)f0.mu_p + wg1 * (double)f0.sig_p;
            const double zs_f = wf0 * (double)f0.mu_p + wf1 * (double)f0.sig_p;
            double sz[4] = {0.0, 0.0, 0.0, 0.0};
            for (int n = act ? gt : N; n < N; n += 256) {
                const FwdPlaneT<R> f = plane_of(n);
                const double dg = wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - zs_g;
                const double df = wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - zs_f;
                sz[0] += dg;
                sz[1] += dg * dg;
                sz[2] += df;
                sz[3] += df * df;
            }
            local_group_sum<4>(sz, red);
            mg = zs_g + sz[0] * a.inv_n;
            mf = zs_f + sz[2] * a.inv_n;
            double vg = (sz[1] - sz[0] * sz[0] * a.inv_n) * a.inv_n, vf = (sz[3] - sz[2] * sz[2] * a.inv_n) * a.inv_n;
            vg = vg > 0.0 ? vg : 0.0;
            vf = vf > 0.0 ? vf : 0.0;
            rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
            rf = (double)__builtin_amdgcn_rsqf((float)(vf + (double)a.eps_bn));
            if (act && gt == 0) {
                const double mom_ = a.momentum, unb = a.unbias_n;
                gg.run_mean[c] = (float)((1.0 - mom_) * q[4] + mom_ * mg);
                gg.run_var[c] = (float)((1.0 - mom_) * q[5] + mom_ * vg * unb);
                if (a.sn_two) {
                    gf.run_mean[c] = (float)((1.0 - mom_) * q[10] + mom_ * mf);
                    gf.run_var[c] = (float)((1.0 - mom_) * q[11] + mom_ * vf * unb);
                }
                if (c == 0) {
                    bump_batches_tracked(gg.nbt);
                    if (a.sn_two) bump_batches_tracked(gf.nbt);
                }
            }
        } else {
            rg = (double)__builtin_amdgcn_rsqf((float)q[5] + a.eps_bn);
            rf = a.sn_two ? (double)__builtin_amdgcn_rsqf((float)q[11] + a.eps_bn) : 1.0;
        }
        if (saved && act && gt == 0) {
            saved[SV_ROWS * P + c] = rg;
            saved[SV_ROWS * P + C + c] = rf;
        }
        __syncthreads();  // every reader of pmu/pm2 of this channel's statistics sums is done before they are replaced
        for (int n = act ? gt : N; n < N; n += 256) {
            const FwdPlaneT<R> f = plane_of(n);
            const double zhg = (wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - mg) * rg;
            const R g = sigmoid_r<R>((R)(q[2] * zhg + q[3]));
            double zhf = 0.0;
            R fg = 1.f;
            if (a.sn_two) {
                zhf = (wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - mf) * rf;
                fg = sigmoid_r<R>((R)(q[8] * zhf + q[9]));
            }
            const FwdCoefs cf = fwd_coefs<R>(a, f, g, fg);
            if (saved) {
                const SvRec p = sv_rec(n, c, N);
                store_fwd_plane<R>(saved, P, p, f, 0);
                saved[sv_at(p, SV_G)] = g;
                saved[sv_at(p, SV_ZH_G)] = zhg;
                saved[sv_at(p, SV_F)] = fg;
                saved[sv_at(p, SV_ZH_F)] = zhf;
                if (a.save_coefs) store_fwd_coefs(saved, p, cf);
            }
            const int pl = n * CG + cc;
            pmu[pl] = cf.a_in;  // SelfNorm alone: y = a_in * x + b_in  (xr = 0)
            pm2[pl] = cf.b_in;
        }
    }
    __syncthreads();
    LSTAMP();

    // ---- apply from LDS, the only write of y
    local_emit<T, W, LB>(la, y, c0, [&](int pl, int idx) {
        float t = fmaf(pmu[pl], to_float(img[idx]), pm2[pl]);
        if (EPI) t = relu ? fmaxf(t, 0.f) : t;
        return from_float<T>(t);
    });
    LSTAMP();
#ifdef CNSN_LPROF
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
        printf("[local fwd] block %d: stage %lld stats %lld gates %lld emit %lld (x10ns)\n", (int)blockIdx.x, t_[1] - t_[0],
               t_[2] - t_[1], t_[3] - t_[2], t_[4] - t_[3]);
#endif
}

// ================================================================================================
// backward
// ================================================================================================
template <typename T, int W, bool EPI, int LB>
__global__ __launch_bounds__(LB) void local_bwd_kernel(LocalArgs la, const T* __restrict__ gy,
                                                           const T* __restrict__ x, const T* __restrict__ addend,
                                                           T* __restrict__ dx, GateDev gg, GateDev gf, GateGradDev dgr,
                                                           GateGradDev dfr, const double* __restrict__ saved, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = la.mid;
    const int N = a.N, C = a.C, M = a.M, CG = la.CG, planes = la.planes;
    const int c0 = blockIdx.x * CG;
    const size_t img_bytes = local_align((size_t)planes * M * sizeof(T));
    T* gimg = (T*)smem;
    T* ximg = (T*)(smem + img_bytes);
    float* ps1 = (float*)(smem + 2 * img_bytes);  // [planes] sum G        -> later cG
    float* ps2 = ps1 + planes;                    // [planes] sum G*(x-mu) -> later cX
    float* pxr = ps2 + planes;                    // [planes] xr
    float* pc0 = pxr + planes;                    // [planes] c0
    double* pdt = (double*)((char*)ps1 + local_align((size_t)4 * planes * 4));  // [LB/256][2][N] dt of the channels in work
    double* red = pdt + (size_t)(LB / 256) * 2 * N;
    const size_t P = (size_t)N * C;

    local_stage<T, W, false, LB>(la, gy, nullptr, c0, smem);
    if (EPI && addend)
        local_stage<T, W, true, LB>(la, x, addend, c0, smem + img_bytes);
    else
        local_stage<T, W, false, LB>(la, x, nullptr, c0, smem + img_bytes);
    __syncthreads();

    // ---- ReLU mask (forward affine re-evaluated with the saved coefficients) and per-plane sums
    {
        const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
        for (int pl = grp; pl < planes; pl += LB / 16) {
            const int n = pl / CG, cc = pl - n * CG;
            const SvRec p = sv_rec(n, c0 + cc, N);
            const float si = (float)saved[sv_at(p, SV_MU_C)];
            float fa = 0.f, fb = 0.f;
            if (EPI && relu) {
                fa = (float)saved[sv_at(p, SV_FC0 + FC_A_IN)];
                fb = (float)saved[sv_at(p, SV_FC0 + FC_B_IN)];
            }
            T* gp = gimg + (size_t)pl * M;
            const T* xp = ximg + (size_t)pl * M;
            float s1 = 0.f, s2 = 0.f;
            for (int e = l16; e < M; e += 16) {
                const float X = to_float(xp[e]);
                float G = to_float(gp[e]);
                if (EPI && relu) {
                    if (!relu_open<T>(fmaf(fa, X, fb))) {
                        G = 0.f;
                        gp[e] = from_float<T>(0.f);
                    }
                }
                s1 += G;
                s2 = fmaf(G, X - si, s2);
            }
            s1 = row16_sum(s1);
            s2 = row16_sum(s2);
            if (l16 == 0) {
                ps1[pl] = s1;
                ps2[pl] = s2;
            }
        }
    }
    __syncthreads();

    // ---- gate / BatchNorm backward per channel; coefficients of dx per plane
    using R = float;
    constexpr int SUB = LB / 256;  // 256-thread groups, one channel each at a time
    const int gidx = threadIdx.x >> 8, gt = threadIdx.x & 255;
    double* gdt = pdt + (size_t)gidx * 2 * N;
    for (int cc0 = 0; cc0 < CG; cc0 += SUB) {
        const bool act = cc0 + gidx < CG;
        const int cc = act ? cc0 + gidx : 0;
        const int c = c0 + cc;
        auto rec = [&](int n, int row) { return saved[sv_at(sv_rec(n, c, N), row)]; };
        auto sums_of = [&](int n) {
            const int pl = n * CG + cc;
            return fix_sums<R>(a, ps1[pl], ps2[pl], 0.f, 0.f, rec(n, SV_MU_C), 0.0);
        };
        double s4[4] = {0, 0, 0, 0};
        for (int n = act ? gt : N; n < N; n += 256) {
            R dtg, dtf;
            const R mu = (R)rec(n, SV_MU_C);
            gate_dt<R>(a, sums_of(n), R(1), mu, R(0), (R)rec(n, SV_MU_P), (R)rec(n, SV_G), (R)rec(n, SV_F), dtg, dtf);
            s4[0] += (double)dtg;
            s4[1] += (double)dtg * rec(n, SV_ZH_G);
            s4[2] += (double)dtf;
            s4[3] += (double)dtf * rec(n, SV_ZH_F);
            gdt[n] = dtg;
            gdt[N + n] = dtf;
        }
        local_group_sum<4>(s4, red);
        BnBwd b{};
        b.s_dt_g = s4[0];
        b.s_dtz_g = s4[1];
        b.s_dt_f = s4[2];
        b.s_dtz_f = s4[3];
        b.wg0 = gg.w[2 * c];
        b.wg1 = gg.w[2 * c + 1];
        b.kg = (double)gg.gamma[c] * saved[SV_ROWS * P + c];
        if (a.sn_two) {
            b.wf0 = gf.w[2 * c];
            b.wf1 = gf.w[2 * c + 1];
            b.kf = (double)gf.gamma[c] * saved[SV_ROWS * P + C + c];
        }
        double sw[4] = {0, 0, 0, 0};
        for (int n = act ? gt : N; n < N; n += 256) {
            const R mu = (R)rec(n, SV_MU_C), mu_p = (R)rec(n, SV_MU_P), sig_p = (R)rec(n, SV_SIG_P);
            const R g = (R)rec(n, SV_G), f = (R)rec(n, SV_F);
            const BwdPlaneT<R> o = bwd_plane<R>(a, b, sums_of(n), gdt[n], gdt[N + n], rec(n, SV_ZH_G), rec(n, SV_ZH_F), g, f,
                                                R(1), R(1), mu, mu_p, sig_p, R(1), R(0));
            sw[0] += (double)o.dz_g * (double)mu_p;
            sw[1] += (double)o.dz_g * (double)sig_p;
            sw[2] += (double)o.dz_f * (double)mu_p;
            sw[3] += (double)o.dz_f * (double)sig_p;
            const BwdCoefs k = bwd_coefs<R>(a, o, R(0), R(0), g, R(1), mu, mu_p, rec(n, SV_MU_C), R(1), rec(n, SV_MU_C), R(1));
            const int pl = n * CG + cc;
            ps1[pl] = k.cG_in;
            ps2[pl] = k.cX_in;
            pxr[pl] = k.xr_in;
            pc0[pl] = k.c0_in;
        }
        local_group_sum<4>(sw, red);
        if (act && gt == 0) {
            dgr.dgamma[c] = (float)s4[1];
            dgr.dbeta[c] = (float)s4[0];
            dgr.dw[2 * c] = (float)sw[0];
            dgr.dw[2 * c + 1] = (float)sw[1];
            if (a.sn_two) {
                dfr.dgamma[c] = (float)s4[3];
                dfr.dbeta[c] = (float)s4[2];
                dfr.dw[2 * c] = (float)sw[2];
                dfr.dw[2 * c + 1] = (float)sw[3];
            }
        }
    }
    __syncthreads();

    // ---- dx from LDS, the only write
    local_emit<T, W, LB>(la, dx, c0, [&](int pl, int idx) {
        const float G = to_float(gimg[idx]), X = to_float(ximg[idx]);
        return from_float<T>(fmaf(ps1[pl], G, fmaf(ps2[pl], X - pxr[pl], pc0[pl])));
    });
}

}  // namespace cnsn
