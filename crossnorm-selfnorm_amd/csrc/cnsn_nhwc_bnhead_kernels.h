// Channels-last single-launch kernels WITH THE BLOCK'S LAST BatchNorm2d IN FRONT (round 6):
//     y = act( SelfNorm( BatchNorm2d(conv_out) + identity ) )
// — the tail of a ResNet bottleneck, `out = self.bn3(out); out += identity; out = self.cnsn(out); out = self.relu(out)`
// (models/imagenet/resnet_cnsn.py:108-122, pos='post') — in ONE persistent launch per direction.
//
// Why: in the channels-last ResNet-50 step (BASELINE config 3) the 53 BatchNorm2d layers are a third of the GPU time
// (MIOpen: 6 ms forward + 9 ms backward of a 44 ms step, profiles/r06e_resnet50_channels_last_step_kernel_stats.csv), and the
// 16 bn3 layers — the block's widest tensors — are half of that.  Un-fused, a block's tail moves 18 tensor passes per step:
// BatchNorm2d 3 + 5 (statistics, normalise; two reductions, apply), the op 5 + 5 (cnsn_nhwc_fused_kernels.h).  Here 13:
//   forward   A  read conv_out, identity: per-plane sums of c, c^2, b, b^2, c*b (about the plane's first pixel)
//             B  BatchNorm2d's batch statistics from the plane sums (Chan merge over the batch) -> alpha, beta per channel;
//                the statistics of X = alpha*c + beta + b per plane BY ALGEBRA (mean, M2 from the five sums) -> SelfNorm's gate
//             C  read conv_out, identity again: x = T(alpha*c + beta), X = T(x + b), y = act(g * X)          (2 + 2 + 1 passes)
//   backward  A' read grad_y, conv_out, identity: X recomputed; per-plane sums of G', G'*(X - mu), G'*(c - m)
//             B' SelfNorm's backward per plane -> dX = g*G' + cX*(X - mu) + c0; BatchNorm2d's two channel sums of dX and
//                dX*(c - m) follow from the plane sums and two more numbers per plane the forward kept -> e0, e1 per channel
//             C' read the three tensors again: write dX (the identity branch's gradient) and
//                d conv_out = alpha*dX + e0 + e1*(c - m)                                                    (3 + 3 + 2 passes)
// Nothing but conv_out and the identity is saved for the backward (what BatchNorm2d and the add would have saved anyway);
// neither bn3's output nor the sum is ever written.
//
// Numerics.  The reference's statistics are those of the tensor AFTER two roundings to T (bn3's output, the in-place add); here
// they follow from the un-rounded sums — the difference is rounding noise of zero mean (2^-9 relative per element in bf16,
// nothing in fp32), far inside north_star's tolerances (tests/test_gpu_bn_block.py against torch's own composition).  The
// element-wise values (x, X, y, the ReLU mask, dX) are rounded exactly where the un-fused sequence rounds them.
// Training mode only (batch statistics in both normalisations): anything else runs the un-fused sequence.
//
// TWO: the identity is itself `BatchNorm2d(conv)` — the block's `downsample` (resnet_cnsn.py:99-100: a 1x1 convolution and a
// BatchNorm2d on the skip path, the first block of every stage).  X = bn3(c1) + bn_d(c2) is affine in BOTH convolution outputs
// per channel, so the same five plane sums give both BatchNorm2d's batch statistics and the statistics of X; the backward writes
// the gradients of the two convolution outputs where it wrote those of conv_out and the identity.  Same 13 passes; the
// downsample's BatchNorm2d costs nothing any more.
#pragma once
#include "cnsn_nhwc_fused_kernels.h"

#ifndef CNSN_BNHEAD_UA
#define CNSN_BNHEAD_UA 2  // pixels of a thread's column in flight in the forward's statistics phase (two tensors each)
#endif
#ifndef CNSN_BNHEAD_WG_PER_CU
#define CNSN_BNHEAD_WG_PER_CU 3  // (168 VGPRs: the backward's apply phase holds nine coefficients per channel of a lane's vector)
#endif

namespace cnsn {

enum BnSlimRow { SLB_E = SL_ROWS, SLB_DM, SLB_E2, SLB_DM2, SLB_ROWS };  // + sum (X - mean)(c - mean_c), mean_c - m per plane, for conv_out and (TWO) the skip path's convolution
__host__ __device__ inline size_t bn_slim_floats(size_t P, int C) { return (size_t)SLB_ROWS * P + 2 * (size_t)C; }
__host__ __device__ inline double* bn_slim_rstd(float* slim, size_t P) { return reinterpret_cast<double*>(slim + (size_t)SLB_ROWS * P); }
__host__ __device__ inline const double* bn_slim_rstd(const float* slim, size_t P) {
    return reinterpret_cast<const double*>(slim + (size_t)SLB_ROWS * P);
}
enum BnStatRow { BS_MEAN = 0, BS_RSTD, BS_ALPHA, BS_BETA, BS_ROWS };  // float (4, C) per BatchNorm2d: what the backward re-evaluates x with; (TWO) the skip path's rows follow: (8, C)

struct BnHeadDev {
    const float* weight;  // (C)
    const float* bias;    // (C)
    float* run_mean;      // (C) updated by the forward
    float* run_var;       // (C) (takes the unbiased batch variance, like torch)
    long long* nbt;       // num_batches_tracked or null
    float eps, momentum;
};

struct NhwcBnArgs {
    NhwcFusedArgs f;   // geometry, SelfNorm's side arrays (f.part: [S][NTOT][P]), the barrier; f.kshift: the conv output's shift
    BnHeadDev bn;
    BnHeadDev bn2;           // (TWO) the skip path's BatchNorm2d
    double inv_r, unbias_r;  // 1 / (N*M), N*M / (N*M - 1)
    float* kshift_b;         // [P] shift of the identity's sums
    float* chan;             // backward: [2][C] e0, e1 (phase B' -> C'); (TWO) [4][C]: + those of the skip path's convolution
    float* bn_stats;         // [BS_ROWS][C]: forward writes, backward reads
    float* dbn_w;            // backward: (C) gradients of BatchNorm2d's weight / bias
    float* dbn_b;
    float* dbn2_w;           // (TWO) ... of the skip path's
    float* dbn2_b;
};

// NK of NTOT accumulators of a tile -> part rows (s*NTOT + k0 + k): the rows of the block added in a fixed order by ALL threads,
// 16-byte write-through stores (the COH branch of nhwc_rows_sum, for a subset of the accumulators: the staging area holds NK)
template <int VEC, int NK, int NTOT>
__device__ __forceinline__ void nhwc_rows_sum_part(const NhwcGeom& g, const NhwcThread<VEC>& t, const float (&acc)[NK][VEC], int k0,
                                                   float* lds, float* __restrict__ part) {
    const int col = (int)threadIdx.x % g.tcb, width = g.tcb * VEC;
    if (t.r < g.rows) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j) lds[((size_t)k * g.rows + t.r) * width + col * VEC + j] = t.active ? acc[k][j] : 0.f;
    }
    __syncthreads();
    const CohBuf pb(part);
    const int total = NK * width, first = (t.vc - col) * VEC;
    for (int ch = (int)threadIdx.x * 4; ch < total; ch += kBlock * 4) {
        const int k = ch / width, off = ch - k * width;
        if (first + off >= g.C) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < g.rows; ++q) {
            const float4 w = *reinterpret_cast<const float4*>(lds + ((size_t)k * g.rows + q) * width + off);
            v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
        }
        pb.st4(((size_t)t.s * NTOT + k0 + k) * g.P + (size_t)t.n * g.C + first + off, v.x, v.y, v.z, v.w);
    }
    __syncthreads();  // (the staging area is the next call's)
}

// the lane's value of a per-channel array in phase B: thread j < GC owns channel c0 + j
template <int GC>
__device__ __forceinline__ double pick(const double (&v)[GC], int j) {
    double r = 0.0;
#pragma unroll
    for (int q = 0; q < GC; ++q)
        if (q == j) r = v[q];
    return r;
}

// X = T(T(alpha*c + beta) + b): bn3's output and the in-place add, rounded where the un-fused sequence rounds them
template <typename T>
__device__ __forceinline__ float bn_sum(float c, float b, float alpha, float beta) {
    const float x = to_float(from_float<T>(fmaf(alpha, c, beta)));
    return sum_t<T>(x, b);
}

// the skip path's value: the identity as it is, or (TWO) T(alpha2*c2 + beta2), the downsample BatchNorm2d's output
template <typename T, bool TWO>
__device__ __forceinline__ float bn_skip(float c2, float alpha2, float beta2) {
    if constexpr (TWO)
        return to_float(from_float<T>(fmaf(alpha2, c2, beta2)));
    else
        return c2;
}

constexpr int kBnGc = 4;   // adjacent channels a workgroup takes in phase B / B'
constexpr int kBnFwd = 5;  // part rows of the forward: sum c', c'^2, b', b'^2, c'*b'
constexpr int kBnBwd = 4;  // ... of the backward: sum G', G'*(X - mu), G'*(c - m) (, TWO: G'*(c2 - m2))

// ================================================================================================
// forward
// ================================================================================================
template <typename T, int VEC, bool KEEP, bool TWO>
__global__ __launch_bounds__(kBlock, CNSN_BNHEAD_WG_PER_CU) void nhwc_bnhead_fwd_kernel(NhwcBnArgs a, const T* __restrict__ cv,
                                                                                         const T* __restrict__ idt, T* __restrict__ y,
                                                                                         GateDev gg) {
    extern __shared__ float lds[];
    __shared__ double red[4 * kBnGc];
    __shared__ int bar_flag;
    constexpr int GC = kBnGc;
    const NhwcGeom& g = a.f.g;

    // ---- A: five partial sums of every tile
    for (int tile = blockIdx.x; tile < a.f.ntiles; tile += gridDim.x) {
        const NhwcThread<VEC> t(g, tile);
        float Kc[VEC], Kb[VEC], acc[3][VEC], acc2[2][VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) Kc[j] = Kb[j] = acc[0][j] = acc[1][j] = acc[2][j] = acc2[0][j] = acc2[1][j] = 0.f;
        if (t.active) {
            const size_t o = t.elem(g, 0);
            const Vec<T, VEC> c0 = load_vec<T, VEC>(cv + o), b0 = load_vec<T, VEC>(idt + o);
#pragma unroll
            for (int j = 0; j < VEC; ++j) Kc[j] = to_float(c0.v[j]), Kb[j] = to_float(b0.v[j]);
            auto eat = [&](const Vec<T, VEC>& vc, const Vec<T, VEC>& vb) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float dc = to_float(vc.v[j]) - Kc[j], db = to_float(vb.v[j]) - Kb[j];
                    acc[0][j] += dc;
                    acc[1][j] = fmaf(dc, dc, acc[1][j]);
                    acc[2][j] += db;
                    acc2[0][j] = fmaf(db, db, acc2[0][j]);
                    acc2[1][j] = fmaf(dc, db, acc2[1][j]);
                }
            };
            constexpr int U = CNSN_BNHEAD_UA;
            int p = t.p0 + t.r;
            for (; p + (U - 1) * g.rows < t.p1; p += U * g.rows) {
                Vec<T, VEC> vc[U], vb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const size_t e = t.elem(g, p + u * g.rows);
                    vc[u] = nhwc_ld<T, VEC, !KEEP>(cv + e);
                    vb[u] = nhwc_ld<T, VEC, !KEEP>(idt + e);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) eat(vc[u], vb[u]);
            }
            for (; p < t.p1; p += g.rows) {
                const size_t e = t.elem(g, p);
                const Vec<T, VEC> vc = nhwc_ld<T, VEC, !KEEP>(cv + e), vb = nhwc_ld<T, VEC, !KEEP>(idt + e);
                eat(vc, vb);
            }
            if (t.s == 0 && t.r == 0) {
                CohBuf(a.f.kshift).store<VEC>(t.plane0(g), Kc);
                CohBuf(a.kshift_b).store<VEC>(t.plane0(g), Kb);
            }
        }
        nhwc_rows_sum_part<VEC, 3, kBnFwd>(g, t, acc, 0, lds, a.f.part);
        nhwc_rows_sum_part<VEC, 2, kBnFwd>(g, t, acc2, 3, lds, a.f.part);
    }
    if (!grid_barrier(a.f.bar, 1, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.f.ntiles, y);
        return;
    }

    // ---- B: BatchNorm2d's batch statistics, the statistics of X by algebra, SelfNorm's gate; thread n = instance n (N <= 256)
    for (int slot = blockIdx.x; slot < a.f.ngroups; slot += gridDim.x) {
        const int grp = phase_b_group(slot, a.f.ngroups);
        const int c0 = grp * GC, n = threadIdx.x;
        const bool live = n < g.N;
        const size_t p0 = (size_t)(live ? n : 0) * g.C + c0;
        double Sc[GC], Qc[GC], Sb[GC], Qb[GC], Pcb[GC], Kc[GC], Kb[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) Sc[j] = Qc[j] = Sb[j] = Qb[j] = Pcb[j] = 0.0;
        const CohBuf pb(a.f.part);
        for (int s = 0; s < g.S; ++s) {
            const size_t base = (size_t)s * kBnFwd * g.P + p0;
            add_group_coh<GC>(pb, base, Sc);
            add_group_coh<GC>(pb, base + g.P, Qc);
            add_group_coh<GC>(pb, base + 2 * g.P, Sb);
            add_group_coh<GC>(pb, base + 3 * g.P, Qb);
            add_group_coh<GC>(pb, base + 4 * g.P, Pcb);
        }
        load_group_coh<GC>(CohBuf(a.f.kshift), p0, Kc);
        load_group_coh<GC>(CohBuf(a.kshift_b), p0, Kb);
        const double M = (double)g.M;
        double mc[GC], M2c[GC], mb[GC], M2b[GC], Ccb[GC], t1[GC], m2d[GC], v2d[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            mc[j] = Kc[j] + Sc[j] / M;
            M2c[j] = Qc[j] - Sc[j] * Sc[j] / M;
            mb[j] = Kb[j] + Sb[j] / M;
            M2b[j] = Qb[j] - Sb[j] * Sb[j] / M;
            Ccb[j] = Pcb[j] - Sc[j] * Sb[j] / M;
            t1[j] = live ? mc[j] : 0.0;
        }
        // BatchNorm2d over (N, H, W): every plane has M elements, so the channel mean is the mean of the plane means
        block_sum_d<GC>(t1, red);
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            m2d[j] = t1[j] * a.f.inv_n;
            const double d = mc[j] - m2d[j];
            t1[j] = live ? M2c[j] + M * d * d : 0.0;
        }
        block_sum_d<GC>(t1, red);
        double al[GC], be[GC], mf[GC], rs2[GC], al2[GC], be2[GC], mf2[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) al2[j] = 1.0, be2[j] = 0.0, mf2[j] = 0.0;  // (not TWO: the identity as it is)
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            v2d[j] = t1[j] * a.inv_r;  // biased: what normalises (torch); the unbiased one goes to running_var
            v2d[j] = v2d[j] > 0.0 ? v2d[j] : 0.0;
            rs2[j] = 1.0 / sqrt(v2d[j] + (double)a.bn.eps);
            const double alpha = (double)a.bn.weight[c0 + j] * rs2[j];
            const float af = (float)alpha, bf = (float)((double)a.bn.bias[c0 + j] - m2d[j] * alpha);
            al[j] = (double)af, be[j] = (double)bf;  // (what phase C and the backward evaluate x with)
            mf[j] = (double)(float)m2d[j];
        }
        if (threadIdx.x < GC) {
            const int j = threadIdx.x, c = c0 + j;
            const double mj = pick<GC>(m2d, j), vj = pick<GC>(v2d, j), mom = (double)a.bn.momentum;
            a.bn.run_mean[c] = (float)((1.0 - mom) * (double)a.bn.run_mean[c] + mom * mj);
            a.bn.run_var[c] = (float)((1.0 - mom) * (double)a.bn.run_var[c] + mom * vj * a.unbias_r);
            if (c == 0) bump_batches_tracked(a.bn.nbt);
            a.bn_stats[(size_t)BS_MEAN * g.C + c] = (float)pick<GC>(mf, j);
            a.bn_stats[(size_t)BS_RSTD * g.C + c] = (float)pick<GC>(rs2, j);
        }
        if (threadIdx.x == 0) {  // (phase C reads these two rows: written through, four channels at a time)
            const CohBuf sb(a.bn_stats);
            sb.st4((size_t)BS_ALPHA * g.C + c0, (float)al[0], (float)al[1], (float)al[2], (float)al[3]);
            sb.st4((size_t)BS_BETA * g.C + c0, (float)be[0], (float)be[1], (float)be[2], (float)be[3]);
        }
        if constexpr (TWO) {  // the skip path's BatchNorm2d, the same way from the second set of plane moments
            double m2b[GC], v2b[GC], rsb[GC];
#pragma unroll
            for (int j = 0; j < GC; ++j) t1[j] = live ? mb[j] : 0.0;
            block_sum_d<GC>(t1, red);
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                m2b[j] = t1[j] * a.f.inv_n;
                const double d = mb[j] - m2b[j];
                t1[j] = live ? M2b[j] + M * d * d : 0.0;
            }
            block_sum_d<GC>(t1, red);
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                v2b[j] = t1[j] * a.inv_r;
                v2b[j] = v2b[j] > 0.0 ? v2b[j] : 0.0;
                rsb[j] = 1.0 / sqrt(v2b[j] + (double)a.bn2.eps);
                const double alpha = (double)a.bn2.weight[c0 + j] * rsb[j];
                const float af = (float)alpha, bf = (float)((double)a.bn2.bias[c0 + j] - m2b[j] * alpha);
                al2[j] = (double)af, be2[j] = (double)bf;
                mf2[j] = (double)(float)m2b[j];
            }
            float* st2 = a.bn_stats + (size_t)BS_ROWS * g.C;
            if (threadIdx.x < GC) {
                const int j = threadIdx.x, c = c0 + j;
                const double mj = pick<GC>(m2b, j), vj = pick<GC>(v2b, j), mom = (double)a.bn2.momentum;
                a.bn2.run_mean[c] = (float)((1.0 - mom) * (double)a.bn2.run_mean[c] + mom * mj);
                a.bn2.run_var[c] = (float)((1.0 - mom) * (double)a.bn2.run_var[c] + mom * vj * a.unbias_r);
                if (c == 0) bump_batches_tracked(a.bn2.nbt);
                st2[(size_t)BS_MEAN * g.C + c] = (float)pick<GC>(mf2, j);
                st2[(size_t)BS_RSTD * g.C + c] = (float)pick<GC>(rsb, j);
            }
            if (threadIdx.x == 0) {
                const CohBuf sb(st2);
                sb.st4((size_t)BS_ALPHA * g.C + c0, (float)al2[0], (float)al2[1], (float)al2[2], (float)al2[3]);
                sb.st4((size_t)BS_BETA * g.C + c0, (float)be2[0], (float)be2[1], (float)be2[2], (float)be2[3]);
            }
        }
        // the statistics of X = alpha*c + beta + b (TWO: b = alpha2*c2 + beta2) of every plane: models/cnsn.py:14,133 on the sum
        // the block forms (:117)
        double mean[GC], sig[GC], z[GC], s1[GC], s2[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            mean[j] = al[j] * mc[j] + be[j] + al2[j] * mb[j] + be2[j];
            const double m2 = al[j] * al[j] * M2c[j] + al2[j] * al2[j] * M2b[j] + 2.0 * al[j] * al2[j] * Ccb[j];
            sig[j] = sqrt((m2 > 0.0 ? m2 : 0.0) / (M - 1.0) + (double)a.f.eps_sn);
            z[j] = (double)gg.w[2 * (c0 + j)] * mean[j] + (double)gg.w[2 * (c0 + j) + 1] * sig[j];
            s1[j] = live ? z[j] : 0.0;
        }
        double mz[GC], rstd[GC];
        block_sum_d<GC>(s1, red);  // BatchNorm1d over the batch (:121,138), training mode
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            mz[j] = s1[j] * a.f.inv_n;
            const double d = z[j] - mz[j];
            s2[j] = live ? d * d : 0.0;
        }
        block_sum_d<GC>(s2, red);
#pragma unroll
        for (int j = 0; j < GC; ++j) rstd[j] = 1.0 / sqrt(s2[j] * a.f.inv_n + (double)a.f.eps_bn);
        if (threadIdx.x < GC) {
            const int j = threadIdx.x, c = c0 + j;
            const double vj = pick<GC>(s2, j) * a.f.inv_n, mj = pick<GC>(mz, j), mom = a.f.momentum;
            gg.run_mean[c] = (float)((1.0 - mom) * gg.run_mean[c] + mom * mj);
            gg.run_var[c] = (float)((1.0 - mom) * gg.run_var[c] + mom * vj * a.f.unbias_n);
            if (c == 0) bump_batches_tracked(gg.nbt);
            if (a.f.slim) bn_slim_rstd(a.f.slim, g.P)[c] = pick<GC>(rstd, j);
        }
        if (live) {
            float o_g[GC], o_zh[GC], o_hi[GC], o_lo[GC], o_sig[GC], o_e[GC], o_dm[GC], o_e2[GC], o_dm2[GC];
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                const double zh = (z[j] - mz[j]) * rstd[j];
                o_g[j] = (float)sigmoid_d((double)gg.gamma[c0 + j] * zh + (double)gg.beta[c0 + j]);
                o_zh[j] = (float)zh;
                o_hi[j] = (float)mean[j];
                o_lo[j] = (float)(mean[j] - (double)o_hi[j]);
                o_sig[j] = (float)sig[j];
                o_e[j] = (float)(al[j] * M2c[j] + al2[j] * Ccb[j]);   // sum over the plane of (X - mean)*(c - mean_c): the backward's BatchNorm2d sums
                o_dm[j] = (float)(mc[j] - mf[j]);
                o_e2[j] = (float)(al2[j] * M2b[j] + al[j] * Ccb[j]);  // (TWO) ... of (X - mean)*(c2 - mean_c2)
                o_dm2[j] = (float)(mb[j] - mf2[j]);
            }
            CohBuf(a.f.gout).store<GC>(p0, o_g);  // (phase C reads it)
            if (a.f.slim) {
                store_group<GC>(a.f.slim + (size_t)SL_MU_HI * g.P + p0, o_hi);
                store_group<GC>(a.f.slim + (size_t)SL_MU_LO * g.P + p0, o_lo);
                store_group<GC>(a.f.slim + (size_t)SL_SIG * g.P + p0, o_sig);
                store_group<GC>(a.f.slim + (size_t)SL_ZH * g.P + p0, o_zh);
                store_group<GC>(a.f.slim + (size_t)SLB_E * g.P + p0, o_e);
                store_group<GC>(a.f.slim + (size_t)SLB_DM * g.P + p0, o_dm);
                if constexpr (TWO) {
                    store_group<GC>(a.f.slim + (size_t)SLB_E2 * g.P + p0, o_e2);
                    store_group<GC>(a.f.slim + (size_t)SLB_DM2 * g.P + p0, o_dm2);
                }
            }
        }
        __syncthreads();  // (red is the next group's)
    }
    if (!grid_barrier(a.f.bar, 2, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.f.ntiles, y);
        return;
    }

    // ---- C: y = act(g * X), X = T(T(alpha*c + beta) + b); the tiles in reverse order
    const int relu = a.f.relu;
    const int mine = a.f.ntiles > (int)blockIdx.x ? (a.f.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x : -1;
    for (int i = mine; i >= 0; --i) {
        const int tile = blockIdx.x + i * gridDim.x;
        const NhwcThread<VEC> t(g, tile);
        if (!t.active) continue;
        float gate[VEC], al[VEC], be[VEC], al2[TWO ? VEC : 1], be2[TWO ? VEC : 1];
        CohBuf(a.f.gout).load<VEC>(t.plane0(g), gate);
        const CohBuf sb(a.bn_stats);
        sb.load<VEC>((size_t)BS_ALPHA * g.C + (size_t)t.vc * VEC, al);
        sb.load<VEC>((size_t)BS_BETA * g.C + (size_t)t.vc * VEC, be);
        if constexpr (TWO) {
            sb.load<VEC>((size_t)(BS_ROWS + BS_ALPHA) * g.C + (size_t)t.vc * VEC, al2);
            sb.load<VEC>((size_t)(BS_ROWS + BS_BETA) * g.C + (size_t)t.vc * VEC, be2);
        }
        auto emit = [&](const Vec<T, VEC>& vc, const Vec<T, VEC>& vb, size_t e) {
            Vec<T, VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float b = bn_skip<T, TWO>(to_float(vb.v[j]), al2[TWO ? j : 0], be2[TWO ? j : 0]);
                const float X = bn_sum<T>(to_float(vc.v[j]), b, al[j], be[j]);
                const float v = gate[j] * X;  // one rounding, like the reference's x * g (:150)
                o.v[j] = from_float<T>(relu ? fmaxf(v, 0.f) : v);
            }
            store_vec_nt<T, VEC>(y + e, o);
        };
        constexpr int U = 2;
        const int cnt = (t.p1 - t.p0 - t.r + g.rows - 1) / g.rows;  // pixels of this thread in the chunk
        int q = cnt - 1;
        for (; q - (U - 1) >= 0; q -= U) {
            Vec<T, VEC> vc[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t e = t.elem(g, t.p0 + t.r + (q - u) * g.rows);
                vc[u] = load_vec_nt<T, VEC>(cv + e);
                vb[u] = load_vec_nt<T, VEC>(idt + e);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) emit(vc[u], vb[u], t.elem(g, t.p0 + t.r + (q - u) * g.rows));
        }
        for (; q >= 0; --q) {
            const size_t e = t.elem(g, t.p0 + t.r + q * g.rows);
            emit(load_vec_nt<T, VEC>(cv + e), load_vec_nt<T, VEC>(idt + e), e);
        }
    }
}

// ================================================================================================
// backward
// ================================================================================================
template <typename T, int VEC, bool KEEP, bool TWO>
__global__ __launch_bounds__(kBlock, TWO ? 2 : CNSN_BNHEAD_WG_PER_CU) void nhwc_bnhead_bwd_kernel(NhwcBnArgs a, const T* __restrict__ gy,
                                                                                         const T* __restrict__ cv,
                                                                                         const T* __restrict__ idt, T* __restrict__ dconv,
                                                                                         T* __restrict__ didt, GateDev gg, GateGradDev dg) {
    extern __shared__ float lds[];
    __shared__ double red[4 * 2 * kBnGc];
    __shared__ int bar_flag;
    constexpr int GC = kBnGc;
    const NhwcGeom& g = a.f.g;
    const float* __restrict__ row_mu = a.f.slim + (size_t)SL_MU_HI * g.P;
    const float* __restrict__ row_g = a.f.slim + (size_t)SL_G * g.P;
    const float* __restrict__ st_m = a.bn_stats + (size_t)BS_MEAN * g.C;
    const float* __restrict__ st_al = a.bn_stats + (size_t)BS_ALPHA * g.C;
    const float* __restrict__ st_be = a.bn_stats + (size_t)BS_BETA * g.C;
    const float* __restrict__ st2 = a.bn_stats + (size_t)BS_ROWS * g.C;  // (TWO) the skip path's rows
    const int relu = a.f.relu;

    // ---- A': per-(n, c) sums of G', G'*(X - float(mean)), G'*(c - float(m)) over a pixel chunk
    for (int tile = blockIdx.x; tile < a.f.ntiles; tile += gridDim.x) {
        const NhwcThread<VEC> t(g, tile);
        float acc[3][VEC], acc4[1][VEC], mu[VEC], gate[VEC], al[VEC], be[VEC], mch[VEC], al2[TWO ? VEC : 1], be2[TWO ? VEC : 1],
            m2ch[TWO ? VEC : 1];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = acc[2][j] = acc4[0][j] = mu[j] = gate[j] = al[j] = be[j] = mch[j] = 0.f;
#pragma unroll
        for (int j = 0; j < (TWO ? VEC : 1); ++j) al2[j] = be2[j] = m2ch[j] = 0.f;
        if (t.active) {
            const size_t pl = t.plane0(g), ch = (size_t)t.vc * VEC;
            load_planes<VEC>(row_mu + pl, mu);
            load_planes<VEC>(row_g + pl, gate);
            load_planes<VEC>(st_al + ch, al);
            load_planes<VEC>(st_be + ch, be);
            load_planes<VEC>(st_m + ch, mch);
            if constexpr (TWO) {
                load_planes<VEC>(st2 + (size_t)BS_ALPHA * g.C + ch, al2);
                load_planes<VEC>(st2 + (size_t)BS_BETA * g.C + ch, be2);
                load_planes<VEC>(st2 + (size_t)BS_MEAN * g.C + ch, m2ch);
            }
            auto eat = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vc, const Vec<T, VEC>& vb) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float c = to_float(vc.v[j]), c2 = to_float(vb.v[j]);
                    const float X = bn_sum<T>(c, bn_skip<T, TWO>(c2, al2[TWO ? j : 0], be2[TWO ? j : 0]), al[j], be[j]);
                    float G = to_float(vg.v[j]);
                    if (relu) G = relu_open<T>(gate[j] * X) ? G : 0.f;
                    acc[0][j] += G;
                    acc[1][j] = fmaf(G, X - mu[j], acc[1][j]);
                    acc[2][j] = fmaf(G, c - mch[j], acc[2][j]);
                    if constexpr (TWO) acc4[0][j] = fmaf(G, c2 - m2ch[j], acc4[0][j]);
                }
            };
            int p = t.p0 + t.r;
            for (; p + g.rows < t.p1; p += 2 * g.rows) {
                Vec<T, VEC> vg[2], vc[2], vb[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const size_t e = t.elem(g, p + u * g.rows);
                    vg[u] = nhwc_ld<T, VEC, !KEEP>(gy + e);
                    vc[u] = nhwc_ld<T, VEC, !KEEP>(cv + e);
                    vb[u] = nhwc_ld<T, VEC, !KEEP>(idt + e);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) eat(vg[u], vc[u], vb[u]);
            }
            for (; p < t.p1; p += g.rows) {
                const size_t e = t.elem(g, p);
                eat(nhwc_ld<T, VEC, !KEEP>(gy + e), nhwc_ld<T, VEC, !KEEP>(cv + e), nhwc_ld<T, VEC, !KEEP>(idt + e));
            }
        }
        nhwc_rows_sum_part<VEC, 3, kBnBwd>(g, t, acc, 0, lds, a.f.part);
        if constexpr (TWO) nhwc_rows_sum_part<VEC, 1, kBnBwd>(g, t, acc4, 3, lds, a.f.part);
    }
    if (!grid_barrier(a.f.bar, 1, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.f.ntiles, dconv);
        return;
    }

    // ---- B': SelfNorm's gate / BatchNorm1d backward per channel (cnsn_nhwc_fused_kernels.h), then BatchNorm2d's two channel sums
    for (int slot = blockIdx.x; slot < a.f.ngroups; slot += gridDim.x) {
        const int grp = phase_b_group(slot, a.f.ngroups);
        const int c0 = grp * GC, n = threadIdx.x;
        const bool live = n < g.N;
        const size_t p0 = (size_t)(live ? n : 0) * g.C + c0;
        double S1[GC], S2[GC], S3[GC], S4[GC], mu[GC], lo[GC], gate[GC], zh[GC], sig[GC], E[GC], DM[GC], E2[GC], DM2[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) S1[j] = S2[j] = S3[j] = S4[j] = E2[j] = DM2[j] = 0.0;
        const CohBuf pb(a.f.part);
        for (int s = 0; s < g.S; ++s) {
            const size_t base = (size_t)s * kBnBwd * g.P + p0;
            add_group_coh<GC>(pb, base, S1);
            add_group_coh<GC>(pb, base + g.P, S2);
            add_group_coh<GC>(pb, base + 2 * g.P, S3);
            if constexpr (TWO) add_group_coh<GC>(pb, base + 3 * g.P, S4);
        }
        load_group<GC>(a.f.slim + (size_t)SL_MU_HI * g.P + p0, mu);
        load_group<GC>(a.f.slim + (size_t)SL_MU_LO * g.P + p0, lo);
        load_group<GC>(a.f.slim + (size_t)SL_G * g.P + p0, gate);
        load_group<GC>(a.f.slim + (size_t)SL_ZH * g.P + p0, zh);
        load_group<GC>(a.f.slim + (size_t)SL_SIG * g.P + p0, sig);
        load_group<GC>(a.f.slim + (size_t)SLB_E * g.P + p0, E);
        load_group<GC>(a.f.slim + (size_t)SLB_DM * g.P + p0, DM);
        if constexpr (TWO) {
            load_group<GC>(a.f.slim + (size_t)SLB_E2 * g.P + p0, E2);
            load_group<GC>(a.f.slim + (size_t)SLB_DM2 * g.P + p0, DM2);
        }
        double dt[GC], acc[2 * GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            const double s2c = S2[j] - lo[j] * S1[j];  // about the exact mean: sum G'*(X - mean)
            const double mex = mu[j] + lo[j];
            dt[j] = (s2c + mex * S1[j]) * gate[j] * (1.0 - gate[j]);  // dL/dg = sum G'*X, through the sigmoid
            acc[j] = live ? dt[j] : 0.0;
            acc[GC + j] = live ? dt[j] * zh[j] : 0.0;
        }
        block_sum_d<2 * GC>(acc, red);
        if (threadIdx.x < GC) {
            double sd = 0.0, sdz = 0.0;
#pragma unroll
            for (int q = 0; q < GC; ++q)
                if (q == (int)threadIdx.x) sd = acc[q], sdz = acc[GC + q];
            dg.dgamma[c0 + threadIdx.x] = (float)sdz;
            dg.dbeta[c0 + threadIdx.x] = (float)sd;
        }
        const double M = (double)g.M;
        double dz[GC], cx[GC], cz[GC];
        float o_cx[GC], o_c0[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            const double kg = (double)gg.gamma[c0 + j] * bn_slim_rstd(a.f.slim, g.P)[c0 + j];
            dz[j] = kg * (dt[j] - acc[j] * a.f.inv_n - zh[j] * acc[GC + j] * a.f.inv_n);
            const double dmu = dz[j] * (double)gg.w[2 * (c0 + j)], dsig = dz[j] * (double)gg.w[2 * (c0 + j) + 1];
            cx[j] = dsig / (sig[j] * (M - 1.0));
            cz[j] = dmu / M - cx[j] * lo[j];  // dX = g*G' + cX*(X - float(mean)) + c0
            o_cx[j] = (float)cx[j];
            o_c0[j] = (float)cz[j];
        }
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            acc[j] = live ? dz[j] * (mu[j] + lo[j]) : 0.0;
            acc[GC + j] = live ? dz[j] * sig[j] : 0.0;
        }
        block_sum_d<2 * GC>(acc, red);
        if (threadIdx.x < GC) {
            double w0 = 0.0, w1 = 0.0;
#pragma unroll
            for (int q = 0; q < GC; ++q)
                if (q == (int)threadIdx.x) w0 = acc[q], w1 = acc[GC + q];
            dg.dw[2 * (c0 + threadIdx.x)] = (float)w0;
            dg.dw[2 * (c0 + threadIdx.x) + 1] = (float)w1;
        }
        // BatchNorm2d: D1 = sum over (n, pixels) of dX, D2 = sum of dX*(c - m) — per plane from the sums (float coefficients, as
        // phase C' applies them): sum(X - mu_f) = M*lo, sum(c - m_f) = M*DM, sum (X - mu_f)(c - m_f) = E + lo*M*DM
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            const double fx = (double)o_cx[j], f0 = (double)o_c0[j];
            acc[j] = live ? gate[j] * S1[j] + fx * M * lo[j] + f0 * M : 0.0;
            acc[GC + j] = live ? gate[j] * S3[j] + fx * (E[j] + lo[j] * M * DM[j]) + f0 * M * DM[j] : 0.0;
        }
        block_sum_d<2 * GC>(acc, red);
        if (threadIdx.x < GC) {
            const int c = c0 + threadIdx.x;
            double D1 = 0.0, D2 = 0.0;
#pragma unroll
            for (int q = 0; q < GC; ++q)
                if (q == (int)threadIdx.x) D1 = acc[q], D2 = acc[GC + q];
            const double rs = (double)a.bn_stats[(size_t)BS_RSTD * g.C + c], al = (double)a.bn_stats[(size_t)BS_ALPHA * g.C + c];
            a.dbn_b[c] = (float)D1;
            a.dbn_w[c] = (float)(rs * D2);
            // d conv_out = alpha * (dX - D1/R - (c - m)*rstd^2*D2/R)
            const float e0 = (float)(-al * D1 * a.inv_r), e1 = (float)(-al * rs * rs * D2 * a.inv_r);
            __hip_atomic_store(a.chan + c, e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // (phase C' reads them)
            __hip_atomic_store(a.chan + g.C + c, e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (TWO) {  // the skip path's BatchNorm2d: the same D1, D2 against ITS convolution output
            double a2[GC];
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                const double fx = (double)o_cx[j], f0 = (double)o_c0[j];
                a2[j] = live ? gate[j] * S4[j] + fx * (E2[j] + lo[j] * M * DM2[j]) + f0 * M * DM2[j] : 0.0;
            }
            // (D1 is the same sum: kept from the block sum above by the threads that own a channel)
            double D1k = 0.0;
            if (threadIdx.x < GC) {
#pragma unroll
                for (int q = 0; q < GC; ++q)
                    if (q == (int)threadIdx.x) D1k = acc[q];
            }
            block_sum_d<GC>(a2, red);
            if (threadIdx.x < GC) {
                const int c = c0 + threadIdx.x;
                const double D2b = pick<GC>(a2, (int)threadIdx.x);
                const float* s2 = a.bn_stats + (size_t)BS_ROWS * g.C;
                const double rs = (double)s2[(size_t)BS_RSTD * g.C + c], al = (double)s2[(size_t)BS_ALPHA * g.C + c];
                a.dbn2_b[c] = (float)D1k;
                a.dbn2_w[c] = (float)(rs * D2b);
                const float e0 = (float)(-al * D1k * a.inv_r), e1 = (float)(-al * rs * rs * D2b * a.inv_r);
                __hip_atomic_store(a.chan + 2 * (size_t)g.C + c, e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.chan + 3 * (size_t)g.C + c, e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (live) {
            const CohBuf cb(a.f.coefb);  // (phase C' reads them)
            cb.store<GC>(p0, o_cx);
            cb.store<GC>(g.P + p0, o_c0);
        }
        __syncthreads();
    }
    if (!grid_barrier(a.f.bar, 2, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.f.ntiles, dconv);
        return;
    }

    // ---- C': dX = g*G' + cX*(X - float(mean)) + c0 (the identity's gradient), d conv_out = alpha*dX + e0 + e1*(c - float(m))
    const int mine = a.f.ntiles > (int)blockIdx.x ? (a.f.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x : -1;
    for (int i = mine; i >= 0; --i) {
        const int tile = blockIdx.x + i * gridDim.x;
        const NhwcThread<VEC> t(g, tile);
        if (!t.active) continue;
        const size_t pl = t.plane0(g), ch = (size_t)t.vc * VEC;
        float cG[VEC], cX[VEC], xr[VEC], c0[VEC], al[VEC], be[VEC], mch[VEC], e0[VEC], e1[VEC];
        float al2[TWO ? VEC : 1], be2[TWO ? VEC : 1], m2ch[TWO ? VEC : 1], e0b[TWO ? VEC : 1], e1b[TWO ? VEC : 1];
        load_planes<VEC>(row_g + pl, cG);
        load_planes<VEC>(row_mu + pl, xr);
        const CohBuf cb(a.f.coefb), eb(a.chan);
        cb.load<VEC>(pl, cX);
        cb.load<VEC>(g.P + pl, c0);
        load_planes<VEC>(st_al + ch, al);
        load_planes<VEC>(st_be + ch, be);
        load_planes<VEC>(st_m + ch, mch);
        eb.load<VEC>(ch, e0);
        eb.load<VEC>((size_t)g.C + ch, e1);
        if constexpr (TWO) {
            load_planes<VEC>(st2 + (size_t)BS_ALPHA * g.C + ch, al2);
            load_planes<VEC>(st2 + (size_t)BS_BETA * g.C + ch, be2);
            load_planes<VEC>(st2 + (size_t)BS_MEAN * g.C + ch, m2ch);
            eb.load<VEC>(2 * (size_t)g.C + ch, e0b);
            eb.load<VEC>(3 * (size_t)g.C + ch, e1b);
        }
        auto emit = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vc, const Vec<T, VEC>& vb, size_t e) {
            Vec<T, VEC> o, od;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float c = to_float(vc.v[j]), c2 = to_float(vb.v[j]);
                const float X = bn_sum<T>(c, bn_skip<T, TWO>(c2, al2[TWO ? j : 0], be2[TWO ? j : 0]), al[j], be[j]);
                float G = to_float(vg.v[j]);
                if (relu) G = relu_open<T>(cG[j] * X) ? G : 0.f;
                const float dX = fmaf(cG[j], G, fmaf(cX[j], X - xr[j], c0[j]));
                if constexpr (TWO)  // the gradient of the skip path's convolution output, through ITS BatchNorm2d
                    od.v[j] = from_float<T>(fmaf(al2[j], dX, fmaf(e1b[j], c2 - m2ch[j], e0b[j])));
                else
                    od.v[j] = from_float<T>(dX);
                o.v[j] = from_float<T>(fmaf(al[j], dX, fmaf(e1[j], c - mch[j], e0[j])));
            }
            store_vec_nt<T, VEC>(dconv + e, o);
            store_vec_nt<T, VEC>(didt + e, od);
        };
        const int cnt = (t.p1 - t.p0 - t.r + g.rows - 1) / g.rows;
        for (int q = cnt - 1; q >= 0; --q) {
            const size_t e = t.elem(g, t.p0 + t.r + q * g.rows);
            emit(load_vec_nt<T, VEC>(gy + e), load_vec_nt<T, VEC>(cv + e), load_vec_nt<T, VEC>(idt + e), e);
        }
    }
}

}  // namespace cnsn
