// Channels-last ("NHWC") SINGLE-LAUNCH kernels (round 6): SelfNorm (+ residual-block epilogue) on a [n][h][w][c] tensor in ONE
// persistent launch per direction — what every SelfNorm site of a channels-last ResNet-50 runs (models/imagenet/resnet_cnsn.py
// :112-122 around models/cnsn.py:130-150; BASELINE config 3).
//
// Round 5's channels-last path was four (forward) and six (backward) launches: a statistics pass, a finishing kernel, the mid
// kernel of the two-pass strategy (one workgroup per 1-8 channels, double precision, 20-27 us), the apply pass — five and seven
// tensor passes with 5-15 us of idle GPU between the launches, 9.5 ms for the 16 sites of a ResNet-50 step against 6.3 ms for
// the NCHW cluster kernels (profiles/r05_nhwc.md).  Here the three phases are ONE grid of co-resident workgroups separated by
// two grid-wide barriers:
//   A  every workgroup walks its tiles (instance, pixel chunk, column block): column sums of X = x [+ addend] about the plane's
//      first pixel; rows merged in LDS, written as partial moments  [chunk][2][plane]
//   -- barrier --
//   B  the per-plane algebra, a workgroup per GC adjacent channels, a thread per instance: partial moments -> mean, std ->
//      z = w0*mean + w1*std -> BatchNorm1d over the batch (block sums in double) -> gate g; writes the apply coefficient
//      (g, one float per plane) and the SLIM `saved` record (below)
//   -- barrier --
//   C  the same tiles again, in REVERSE order (what a workgroup read last is what it reads first: L2, then the 256 MiB
//      Infinity Cache serve the second read of the 51-206 MB tensors of the 7x7 / 14x14 sites): y = act(g * X [+ addend])
// and the backward alike (A': sums of G and G*(X - mean); B': gate / BatchNorm backward, parameter gradients, the two dx
// coefficients per plane; C': dx = g*G + cX*(X - mean) + c0).  In-lane arithmetic only in A and C: a thread owns VEC adjacent
// channels of a pixel, so a plane's statistics are COLUMN sums — no cross-lane reduction at all, where the NCHW kernels need a
// wave reduction per plane.
//
// The barrier (resident_bar_area, cnsn_resident_kernels.h): counters that only ever grow (in the persistent context; the host
// knows what earlier launches left and passes the bases, so nothing is cleared between launches), sharded so that no word
// sees more than an eighth of the grid: a workgroup arrives at the counter of its group (blockIdx % 8), the last arriver of a
// group at the top counter, the last arriver there writes the eight generation words, everybody polls its group's — one lane
// per workgroup, with the library's bounded wait and give-up protocol (the control word flips, the
// pinned host word counts, every workgroup marks what it still owed with NaNs).  No fences: everything that crosses a
// barrier is stored write-through and loaded past the L1 (`sc1`: CohBuf of cnsn_nhwc_kernels.h), every wave waits
// for its stores' acknowledgements before its workgroup arrives.  (The first version fenced — release + acquire, agent scope,
// in every wave — and spent ~200 us per barrier at 1 024 workgroups: 0.41 ms for a 7x7 site whose tensor passes take 0.06.)
//
// `saved`, the SLIM record (declared second contract, tests/test_gpu_saved_contract.py): a channels-last call WITHOUT CrossNorm
// and with ONE gate keeps five floats per plane in plane order (p = n*C + c) — mean as a (hi, lo) float pair, std, gate g,
// normalised pre-activation zh — and C doubles of BatchNorm rstd behind them: 20 bytes per plane where the common record of
// cnsn_layout.h holds 56-96 (7 to 12 doubles), which for a 7x7 bf16 plane of 98 bytes was as much traffic as the tensor itself
// (round-5 review: PMC 2.3 x y).  Both channels-last strategies read and write it (cnsn_nhwc.hip converts for the two-pass
// kernels), so any forward still feeds any backward.
#pragma once
#include "cnsn_nhwc_kernels.h"
#include "cnsn_resident_kernels.h"

#ifndef CNSN_NHWC_GC
#define CNSN_NHWC_GC 8  // adjacent channels a workgroup takes in phase B of the forward
#endif
#ifndef CNSN_NHWC_GC_BWD
#define CNSN_NHWC_GC_BWD 4  // ... of the backward (twice the live values per plane: 8 channels spill 124-164 bytes per lane)
#endif
#ifndef CNSN_NHWC_WG_PER_CU
#define CNSN_NHWC_WG_PER_CU 4  // occupancy the kernels are compiled for (registers: 128 per lane)
#endif

namespace cnsn {

enum SlimRow { SL_MU_HI = 0, SL_MU_LO, SL_SIG, SL_G, SL_ZH, SL_ROWS };
__host__ __device__ inline size_t slim_floats(size_t P, int C) { return (size_t)SL_ROWS * P + 2 * (size_t)C; }
__host__ __device__ inline double* slim_rstd(float* slim, size_t P) { return reinterpret_cast<double*>(slim + (size_t)SL_ROWS * P); }
__host__ __device__ inline const double* slim_rstd(const float* slim, size_t P) {
    return reinterpret_cast<const double*>(slim + (size_t)SL_ROWS * P);
}

struct GridBar {
    char* block;                   // the barrier block: group counters, top counter, generation words (kBarLine apart)
    unsigned long long group_base; // arrivals in every group counter before this launch
    unsigned long long bar_base;   // barriers before this launch
    unsigned* ctl;                 // word 0: the time-out flag every cluster kernel of the library shares
    unsigned ctl_idle;
    unsigned* host_flag;
    long long wait_ticks;
    int fault;                     // tests (CNSN_FAULT_INJECT=1): the last workgroup never arrives at the first barrier
};

struct NhwcFusedArgs {
    NhwcGeom g;
    int ntiles;   // N * S * ncb
    int ngroups;  // C / GC (forward: CNSN_NHWC_GC, backward: CNSN_NHWC_GC_BWD)
    int training, relu, keep;
    float eps_sn, eps_bn, momentum;
    double inv_n, unbias_n;
    float* part;    // [S][2][P] partial sums of the tiles
    float* kshift;  // [P] forward: the shift the sums are taken about (X at the plane's first pixel)
    float* gout;    // [P] forward: the gate as the apply phase reads it (the SL_G row of `slim` when there is one)
    float* coefb;   // [2][P] backward: cX, c0
    float* slim;    // forward: written (may be null); backward: read
    void* sum_out;  // forward, ADD_PRE, SUM kernels: where X = x + addend is kept (phase A writes it, phase C reads it)
    GridBar bar;
};

// k-th barrier of the launch (k = 1, 2); gridDim.x is a multiple of 8.  false: the launch gave up (this workgroup's wait ran
// out, or somebody else's did).
__device__ __forceinline__ bool grid_barrier(const GridBar& b, unsigned k, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every wave: its write-through stores have been acknowledged)
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned grp = blockIdx.x & 7u;
        const unsigned long long want = b.bar_base + k;
        gu64* gen = (gu64*)(b.block + (kBarGen + grp) * kBarLine);
        if (!(b.fault && k == 1 && blockIdx.x + 1 == gridDim.x)) {
            const unsigned long long mine =
                __hip_atomic_fetch_add((gu64*)(b.block + grp * kBarLine), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            if (mine == b.group_base + (unsigned long long)k * (gridDim.x >> 3)) {  // last of the group
                const unsigned long long top =
                    __hip_atomic_fetch_add((gu64*)(b.block + kBarTop * kBarLine), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
                if (top == 8ull * want) {  // last of the grid
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        __hip_atomic_store((gu64*)(b.block + (kBarGen + q) * kBarLine), want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        long long t_start = 0;
        for (unsigned spins = 0;; ++spins) {
            if (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
            __builtin_amdgcn_s_sleep(10);
            if ((spins & 15u) == 15u) {
                const long long now = (long long)wall_clock64();
                if (t_start == 0) t_start = now;
                const unsigned seen = __hip_atomic_load((gu32*)b.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen != b.ctl_idle) {  // somebody gave up already: drain
                    ok = 0;
                    break;
                }
                if (now - t_start > b.wait_ticks) {  // (the protocol of sweep_granules: first to notice flips the word and tells the host)
                    const unsigned prev = __hip_atomic_exchange((gu32*)b.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev == b.ctl_idle && b.host_flag)
                        __hip_atomic_fetch_add(b.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = 0;
                    break;
                }
            }
        }
        *flag = ok;
    }
    __syncthreads();  // (what follows loads the other workgroups' results past the L1: CohBuf)
    return *flag != 0;
}

// a launch that gave up: the first pixel of every column of every tile this workgroup owns reads NaN (loud on the same step)
template <typename T, int VEC>
__device__ __forceinline__ void nhwc_mark_owed(const NhwcGeom& g, int ntiles, T* __restrict__ out) {
    Vec<T, VEC> nanv;
#pragma unroll
    for (int j = 0; j < VEC; ++j) nanv.v[j] = from_float<T>(__builtin_nanf(""));
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const NhwcThread<VEC> t(g, tile);
        if (t.active && t.r == 0) store_vec<T, VEC>(out + t.elem(g, t.p0), nanv);
    }
}

template <typename T, int VEC, bool NT>
__device__ __forceinline__ Vec<T, VEC> nhwc_ld(const T* p) {
    if constexpr (NT)
        return load_vec_nt<T, VEC>(p);
    else
        return load_vec<T, VEC>(p);
}

// Which channel group the workgroup in slot `slot` of phase B takes.  A thread reads GC adjacent floats (16-32 bytes) of a row
// that is C floats long: the 64-byte sector it touches belongs to 2-4 ADJACENT groups.  Workgroups are placed on the XCD
// blockIdx % 8 (observed, not promised), so adjacent groups go to slots 8 apart: the sector is fetched into ONE L2 once
// instead of into several (PMC at 7x7: the backward read 1.6 x the tensor passes' bytes with the identity mapping).
__device__ __forceinline__ int phase_b_group(int slot, int ngroups) {
    if ((ngroups & 7) != 0) return slot;
    const int per = ngroups >> 3;
    return (slot & 7) * per + (slot >> 3);
}

// GC adjacent per-plane floats of another workgroup's making (coherent) / of the previous launch's (plain)
template <int GC>
__device__ __forceinline__ void load_group(const float* __restrict__ p, double (&o)[GC]) {
#pragma unroll
    for (int q = 0; q < GC / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        o[4 * q] = v.x, o[4 * q + 1] = v.y, o[4 * q + 2] = v.z, o[4 * q + 3] = v.w;
    }
}
template <int GC>
__device__ __forceinline__ void load_group_coh(const CohBuf& b, size_t idx, double (&o)[GC]) {
    float f[GC];
    b.load<GC>(idx, f);
#pragma unroll
    for (int j = 0; j < GC; ++j) o[j] = (double)f[j];
}
template <int GC>
__device__ __forceinline__ void add_group_coh(const CohBuf& b, size_t idx, double (&o)[GC]) {
    float f[GC];
    b.load<GC>(idx, f);
#pragma unroll
    for (int j = 0; j < GC; ++j) o[j] += (double)f[j];
}
template <int GC>
__device__ __forceinline__ void store_group(float* __restrict__ p, const float (&o)[GC]) {
#pragma unroll
    for (int q = 0; q < GC / 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}
// ================================================================================================
// forward
// ================================================================================================
// SUM (ADD_PRE only): phase A also writes X = x + addend to a.sum_out (default cache policy), phase C reads that ONE tensor;
// the backward is then the ADD_NONE backward on X (cnsn_epilogue_t.sum_out, ABI 8)
template <typename T, int VEC, int ADD, bool KEEP, bool SUM = false>
__global__ __launch_bounds__(kBlock, CNSN_NHWC_WG_PER_CU) void nhwc_fused_fwd_kernel(NhwcFusedArgs a, const T* __restrict__ x,
                                                                                      const T* __restrict__ addend, T* __restrict__ y,
                                                                                      GateDev gg) {
    static_assert(!SUM || ADD == ADD_PRE, "the kept sum is the PRE add's");
    T* const xsum = SUM ? (T*)a.sum_out : nullptr;  // (written in phase A, read in phase C: no __restrict__)
    extern __shared__ float lds[];
    __shared__ double red[4 * CNSN_NHWC_GC];
    __shared__ int bar_flag;
    constexpr int GC = CNSN_NHWC_GC;
    const NhwcGeom& g = a.g;

    // ---- A: partial moments of every tile
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const NhwcThread<VEC> t(g, tile);
        float K[VEC], acc[2][VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) K[j] = acc[0][j] = acc[1][j] = 0.f;
        if (t.active) {
            const size_t o = t.elem(g, 0);
            const Vec<T, VEC> v0 = load_vec<T, VEC>(x + o);
            Vec<T, VEC> b0 = v0;
            if constexpr (ADD == ADD_PRE) b0 = load_vec<T, VEC>(addend + o);
#pragma unroll
            for (int j = 0; j < VEC; ++j) K[j] = ADD == ADD_PRE ? sum_t<T>(to_float(v0.v[j]), to_float(b0.v[j])) : to_float(v0.v[j]);
            constexpr int U = ADD == ADD_PRE ? 2 : 4;
            auto eat = [&](const Vec<T, VEC>& va, const Vec<T, VEC>& vb, size_t e) {
                Vec<T, VEC> keep;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float X = ADD == ADD_PRE ? sum_t<T>(to_float(va.v[j]), to_float(vb.v[j])) : to_float(va.v[j]);
                    keep.v[j] = from_float<T>(X);  // (exact: sum_t rounds to T)
                    const float d = X - K[j];
                    acc[0][j] += d;
                    acc[1][j] = fmaf(d, d, acc[1][j]);
                }
                if constexpr (SUM) store_vec<T, VEC>(xsum + e, keep);
            };
            int p = t.p0 + t.r;
            for (; p + (U - 1) * g.rows < t.p1; p += U * g.rows) {
                Vec<T, VEC> va[U], vb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const size_t e = t.elem(g, p + u * g.rows);
                    va[u] = nhwc_ld<T, VEC, !KEEP>(x + e);
                    if constexpr (ADD == ADD_PRE) vb[u] = nhwc_ld<T, VEC, !KEEP>(addend + e);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) eat(va[u], ADD == ADD_PRE ? vb[u] : va[u], t.elem(g, p + u * g.rows));
            }
            for (; p < t.p1; p += g.rows) {
                const size_t e = t.elem(g, p);
                const Vec<T, VEC> va = nhwc_ld<T, VEC, !KEEP>(x + e);
                Vec<T, VEC> vb = va;
                if constexpr (ADD == ADD_PRE) vb = nhwc_ld<T, VEC, !KEEP>(addend + e);
                eat(va, vb, e);
            }
            if (t.s == 0 && t.r == 0) CohBuf(a.kshift).store<VEC>(t.plane0(g), K);
        }
        nhwc_rows_sum<VEC, 2, true>(g, t, acc, lds, a.part);
        __syncthreads();  // (lds is the next tile's)
    }
    if (!grid_barrier(a.bar, 1, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.ntiles, y);
        return;
    }

    // ---- B: per-plane algebra, GC adjacent channels per workgroup, thread n = instance n (N <= 256)
    for (int slot = blockIdx.x; slot < a.ngroups; slot += gridDim.x) {
        const int grp = phase_b_group(slot, a.ngroups);
        const int c0 = grp * GC, n = threadIdx.x;
        const bool live = n < g.N;
        const size_t p0 = (size_t)(live ? n : 0) * g.C + c0;
        double s1[GC], s2[GC], mean[GC], sig[GC], z[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) s1[j] = s2[j] = 0.0;
        const CohBuf pb(a.part);
        for (int s = 0; s < g.S; ++s) {
            add_group_coh<GC>(pb, ((size_t)s * 2 + 0) * g.P + p0, s1);
            add_group_coh<GC>(pb, ((size_t)s * 2 + 1) * g.P + p0, s2);
        }
        load_group_coh<GC>(CohBuf(a.kshift), p0, mean);
        const double M = (double)g.M;
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            const double m2 = s2[j] - s1[j] * s1[j] / M;
            mean[j] += s1[j] / M;
            sig[j] = sqrt((m2 > 0.0 ? m2 : 0.0) / (M - 1.0) + (double)a.eps_sn);  // unbiased, eps inside (models/cnsn.py:14,133)
            z[j] = (double)gg.w[2 * (c0 + j)] * mean[j] + (double)gg.w[2 * (c0 + j) + 1] * sig[j];  // Conv1d k=2 groups=C (:137)
            s1[j] = live ? z[j] : 0.0;
        }
        double mz[GC], rstd[GC];
        if (a.training) {  // BatchNorm1d over the batch: biased variance normalises, the unbiased one goes to running_var (:121,138)
            block_sum_d<GC>(s1, red);
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                mz[j] = s1[j] * a.inv_n;
                const double d = z[j] - mz[j];
                s2[j] = live ? d * d : 0.0;
            }
            block_sum_d<GC>(s2, red);
#pragma unroll
            for (int j = 0; j < GC; ++j) rstd[j] = 1.0 / sqrt(s2[j] * a.inv_n + (double)a.eps_bn);
            if (threadIdx.x < GC) {
                const int j = threadIdx.x, c = c0 + j;
                double vj = 0.0, mj = 0.0;
#pragma unroll
                for (int q = 0; q < GC; ++q)
                    if (q == j) vj = s2[q] * a.inv_n, mj = mz[q];
                const double mom = a.momentum;
                gg.run_mean[c] = (float)((1.0 - mom) * gg.run_mean[c] + mom * mj);
                gg.run_var[c] = (float)((1.0 - mom) * gg.run_var[c] + mom * vj * a.unbias_n);
                if (c == 0) bump_batches_tracked(gg.nbt);
            }
        } else {
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                mz[j] = gg.run_mean[c0 + j];
                rstd[j] = 1.0 / sqrt((double)gg.run_var[c0 + j] + (double)a.eps_bn);
            }
        }
        if (a.slim && threadIdx.x < GC) {
            double rj = 0.0;
#pragma unroll
            for (int q = 0; q < GC; ++q)
                if (q == (int)threadIdx.x) rj = rstd[q];
            slim_rstd(a.slim, g.P)[c0 + threadIdx.x] = rj;
        }
        if (live) {
            float o_g[GC], o_zh[GC], o_hi[GC], o_lo[GC], o_sig[GC];
#pragma unroll
            for (int j = 0; j < GC; ++j) {
                const double zh = (z[j] - mz[j]) * rstd[j];
                const double gate = sigmoid_d((double)gg.gamma[c0 + j] * zh + (double)gg.beta[c0 + j]);
                o_g[j] = (float)gate;
                o_zh[j] = (float)zh;
                o_hi[j] = (float)mean[j];
                o_lo[j] = (float)(mean[j] - (double)o_hi[j]);
                o_sig[j] = (float)sig[j];
            }
            CohBuf(a.gout).store<GC>(p0, o_g);  // (phase C reads it)
            if (a.slim) {
                store_group<GC>(a.slim + (size_t)SL_MU_HI * g.P + p0, o_hi);
                store_group<GC>(a.slim + (size_t)SL_MU_LO * g.P + p0, o_lo);
                store_group<GC>(a.slim + (size_t)SL_SIG * g.P + p0, o_sig);
                if (a.gout != a.slim + (size_t)SL_G * g.P) store_group<GC>(a.slim + (size_t)SL_G * g.P + p0, o_g);  // (never: see the host)
                store_group<GC>(a.slim + (size_t)SL_ZH * g.P + p0, o_zh);
            }
        }
        __syncthreads();  // (red is the next group's)
    }
    if (!grid_barrier(a.bar, 2, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.ntiles, y);
        return;
    }

    // ---- C: y = act(g * X [+ addend]); the tiles in reverse order (the second read finds what the first one left in the caches)
    const int mine = a.ntiles > (int)blockIdx.x ? (a.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x : -1;
    for (int i = mine; i >= 0; --i) {
        const int tile = blockIdx.x + i * gridDim.x;
        const NhwcThread<VEC> t(g, tile);
        if (!t.active) continue;
        float gate[VEC];
        CohBuf(a.gout).load<VEC>(t.plane0(g), gate);
        constexpr bool TWO = ADD != ADD_NONE && !SUM;  // a second tensor to read
        constexpr int U = TWO ? 2 : 4;
        const T* const xin = SUM ? (const T*)xsum : x;
        auto emit = [&](const Vec<T, VEC>& va, const Vec<T, VEC>& vb, size_t e) {
            Vec<T, VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float f = to_float(va.v[j]);
                if constexpr (ADD == ADD_PRE && !SUM) f = sum_t<T>(f, to_float(vb.v[j]));
                float v = gate[j] * f;  // one rounding, like the reference's x * g (:150)
                if constexpr (ADD == ADD_POST) v += to_float(vb.v[j]);
                o.v[j] = from_float<T>(a.relu ? fmaxf(v, 0.f) : v);
            }
            store_vec_nt<T, VEC>(y + e, o);
        };
        // (walking the chunk backwards as well)
        const int cnt = (t.p1 - t.p0 - t.r + g.rows - 1) / g.rows;  // pixels of this thread in the chunk
        int q = cnt - 1;
        for (; q - (U - 1) >= 0; q -= U) {
            Vec<T, VEC> va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t e = t.elem(g, t.p0 + t.r + (q - u) * g.rows);
                va[u] = load_vec_nt<T, VEC>(xin + e);
                if constexpr (TWO) vb[u] = load_vec_nt<T, VEC>(addend + e);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) emit(va[u], TWO ? vb[u] : va[u], t.elem(g, t.p0 + t.r + (q - u) * g.rows));
        }
        for (; q >= 0; --q) {
            const size_t e = t.elem(g, t.p0 + t.r + q * g.rows);
            const Vec<T, VEC> va = load_vec_nt<T, VEC>(xin + e);
            Vec<T, VEC> vb = va;
            if constexpr (TWO) vb = load_vec_nt<T, VEC>(addend + e);
            emit(va, vb, e);
        }
    }
}

// ================================================================================================
// backward
// ================================================================================================
template <typename T, int VEC, int ADD, bool KEEP>
__global__ __launch_bounds__(kBlock, CNSN_NHWC_WG_PER_CU) void nhwc_fused_bwd_kernel(NhwcFusedArgs a, const T* __restrict__ gy,
                                                                                      const T* __restrict__ x,
                                                                                      const T* __restrict__ addend, T* __restrict__ dx,
                                                                                      T* __restrict__ d_addend, GateDev gg,
                                                                                      GateGradDev dg) {
    extern __shared__ float lds[];
    __shared__ double red[4 * 2 * CNSN_NHWC_GC_BWD];
    __shared__ int bar_flag;
    constexpr int GC = CNSN_NHWC_GC_BWD;
    const NhwcGeom& g = a.g;
    const float* __restrict__ row_mu = a.slim + (size_t)SL_MU_HI * g.P;
    const float* __restrict__ row_g = a.slim + (size_t)SL_G * g.P;
    const int relu = a.relu;

    // ---- A': per-(n, c) sums of G and G * (X - float(mean)) over a pixel chunk (G masked by the forward's ReLU)
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const NhwcThread<VEC> t(g, tile);
        float acc[2][VEC], si[VEC], gate[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = si[j] = gate[j] = 0.f;
        if (t.active) {
            const size_t pl = t.plane0(g);
            load_planes<VEC>(row_mu + pl, si);
            if (relu) load_planes<VEC>(row_g + pl, gate);
            constexpr int U = CNSN_NHWC_UB;
            auto eat = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, const Vec<T, VEC>& vb) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float G, X;
                    nhwc_pair<T, ADD>(to_float(vg.v[j]), to_float(vx.v[j]), to_float(vb.v[j]), gate[j], 0.f, 0.f, relu, G, X);
                    acc[0][j] += G;
                    acc[1][j] = fmaf(G, X - si[j], acc[1][j]);
                }
            };
            int p = t.p0 + t.r;
            for (; p + (U - 1) * g.rows < t.p1; p += U * g.rows) {
                Vec<T, VEC> vg[U], vx[U], vb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const size_t e = t.elem(g, p + u * g.rows);
                    vg[u] = nhwc_ld<T, VEC, !KEEP>(gy + e);
                    vx[u] = nhwc_ld<T, VEC, !KEEP>(x + e);
                    if constexpr (ADD != ADD_NONE) vb[u] = nhwc_ld<T, VEC, !KEEP>(addend + e);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) eat(vg[u], vx[u], ADD != ADD_NONE ? vb[u] : vx[u]);
            }
            for (; p < t.p1; p += g.rows) {
                const size_t e = t.elem(g, p);
                const Vec<T, VEC> vg = nhwc_ld<T, VEC, !KEEP>(gy + e), vx = nhwc_ld<T, VEC, !KEEP>(x + e);
                Vec<T, VEC> vb = vx;
                if constexpr (ADD != ADD_NONE) vb = nhwc_ld<T, VEC, !KEEP>(addend + e);
                eat(vg, vx, vb);
            }
        }
        nhwc_rows_sum<VEC, 2, true>(g, t, acc, lds, a.part);
        __syncthreads();
    }
    if (!grid_barrier(a.bar, 1, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.ntiles, dx);
        return;
    }

    // ---- B': gate / BatchNorm1d backward per channel (closed form: oracle/closed_form.py, csrc/cnsn_algebra.h with a = a1 = 1)
    for (int slot = blockIdx.x; slot < a.ngroups; slot += gridDim.x) {
        const int grp = phase_b_group(slot, a.ngroups);
        const int c0 = grp * GC, n = threadIdx.x;
        const bool live = n < g.N;
        const size_t p0 = (size_t)(live ? n : 0) * g.C + c0;
        double S1[GC], S2[GC], mu[GC], lo[GC], gate[GC], zh[GC], sig[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) S1[j] = S2[j] = 0.0;
        const CohBuf pb(a.part);
        for (int s = 0; s < g.S; ++s) {
            add_group_coh<GC>(pb, ((size_t)s * 2 + 0) * g.P + p0, S1);
            add_group_coh<GC>(pb, ((size_t)s * 2 + 1) * g.P + p0, S2);
        }
        load_group<GC>(a.slim + (size_t)SL_MU_HI * g.P + p0, mu);
        load_group<GC>(a.slim + (size_t)SL_MU_LO * g.P + p0, lo);
        load_group<GC>(a.slim + (size_t)SL_G * g.P + p0, gate);
        load_group<GC>(a.slim + (size_t)SL_ZH * g.P + p0, zh);
        load_group<GC>(a.slim + (size_t)SL_SIG * g.P + p0, sig);
        double dt[GC], acc[2 * GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            S2[j] -= lo[j] * S1[j];  // pass A' shifted by float(mean): sum G*(X - mean) = S2 + (float(mean) - mean) * S1
            mu[j] += lo[j];
            dt[j] = (S2[j] + mu[j] * S1[j]) * gate[j] * (1.0 - gate[j]);  // dL/dg = sum G*X, through the sigmoid
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
        double dz[GC];
        const double M = (double)g.M;
        float o_cx[GC], o_c0[GC];
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            const double kg = (double)gg.gamma[c0 + j] * slim_rstd(a.slim, g.P)[c0 + j];
            // (cancels catastrophically for small batches / saturated gates: always in double, cnsn_algebra.h::bwd_plane)
            dz[j] = kg * (a.training ? dt[j] - acc[j] * a.inv_n - zh[j] * acc[GC + j] * a.inv_n : dt[j]);
            const double dmu = dz[j] * (double)gg.w[2 * (c0 + j)], dsig = dz[j] * (double)gg.w[2 * (c0 + j) + 1];
            const double k = dsig / (sig[j] * (M - 1.0));
            o_cx[j] = (float)k;
            o_c0[j] = (float)(dmu / M - k * lo[j]);  // evaluated as cX*(X - float(mean)) + c0: the rounding of the reference point folded in
        }
#pragma unroll
        for (int j = 0; j < GC; ++j) {
            acc[j] = live ? dz[j] * mu[j] : 0.0;
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
        if (live) {
            const CohBuf cb(a.coefb);  // (phase C' reads them)
            cb.store<GC>(p0, o_cx);
            cb.store<GC>(g.P + p0, o_c0);
        }
        __syncthreads();
    }
    if (!grid_barrier(a.bar, 2, &bar_flag)) {
        nhwc_mark_owed<T, VEC>(g, a.ntiles, dx);
        return;
    }

    // ---- C': dx = g*G + cX*(X - float(mean)) + c0, tiles in reverse; ADD_POST + ReLU also writes the masked gradient
    const int mine = a.ntiles > (int)blockIdx.x ? (a.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x : -1;
    for (int i = mine; i >= 0; --i) {
        const int tile = blockIdx.x + i * gridDim.x;
        const NhwcThread<VEC> t(g, tile);
        if (!t.active) continue;
        const size_t pl = t.plane0(g);
        float cG[VEC], cX[VEC], xr[VEC], c0[VEC];
        load_planes<VEC>(row_g + pl, cG);
        const CohBuf cb(a.coefb);
        cb.load<VEC>(pl, cX);
        load_planes<VEC>(row_mu + pl, xr);
        cb.load<VEC>(g.P + pl, c0);
        auto emit = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, const Vec<T, VEC>& vb, size_t e) {
            Vec<T, VEC> o, om;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float G, X;
                nhwc_pair<T, ADD>(to_float(vg.v[j]), to_float(vx.v[j]), to_float(vb.v[j]), cG[j], 0.f, 0.f, relu, G, X);
                o.v[j] = from_float<T>(fmaf(cG[j], G, fmaf(cX[j], X - xr[j], c0[j])));
                om.v[j] = from_float<T>(G);
            }
            store_vec_nt<T, VEC>(dx + e, o);
            if constexpr (ADD == ADD_POST) {
                if (d_addend) store_vec_nt<T, VEC>(d_addend + e, om);
            }
        };
        constexpr int U = CNSN_NHWC_UB;
        const int cnt = (t.p1 - t.p0 - t.r + g.rows - 1) / g.rows;
        int q = cnt - 1;
        for (; q - (U - 1) >= 0; q -= U) {
            Vec<T, VEC> vg[U], vx[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t e = t.elem(g, t.p0 + t.r + (q - u) * g.rows);
                vg[u] = load_vec_nt<T, VEC>(gy + e);
                vx[u] = load_vec_nt<T, VEC>(x + e);
                if constexpr (ADD != ADD_NONE) vb[u] = load_vec_nt<T, VEC>(addend + e);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) emit(vg[u], vx[u], ADD != ADD_NONE ? vb[u] : vx[u], t.elem(g, t.p0 + t.r + (q - u) * g.rows));
        }
        for (; q >= 0; --q) {
            const size_t e = t.elem(g, t.p0 + t.r + q * g.rows);
            const Vec<T, VEC> vg = load_vec_nt<T, VEC>(gy + e), vx = load_vec_nt<T, VEC>(x + e);
            Vec<T, VEC> vb = vx;
            if constexpr (ADD != ADD_NONE) vb = load_vec_nt<T, VEC>(addend + e);
            emit(vg, vx, vb, e);
        }
    }
}

// ================================================================================================
// the SLIM record for the two-pass channels-last kernels (cnsn_nhwc.hip): they compute with the common record of cnsn_layout.h
// in their workspace; these two kernels move between the records, so that any forward feeds any backward
// ================================================================================================
__global__ __launch_bounds__(kBlock) void nhwc_slim_from_saved_kernel(const double* __restrict__ saved, int N, int C,
                                                                      float* __restrict__ slim) {
    const size_t P = (size_t)N * C, p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < (size_t)C) slim_rstd(slim, P)[p] = saved[(size_t)SV_ROWS * P + p];
    if (p >= P) return;
    const SvRec ps = sv_rec_of_plane(p, N, C);
    const double mu = saved[sv_at(ps, SV_MU_P)];
    const float hi = (float)mu;
    slim[(size_t)SL_MU_HI * P + p] = hi;
    slim[(size_t)SL_MU_LO * P + p] = (float)(mu - (double)hi);
    slim[(size_t)SL_SIG * P + p] = (float)saved[sv_at(ps, SV_SIG_P)];
    slim[(size_t)SL_G * P + p] = (float)saved[sv_at(ps, SV_G)];
    slim[(size_t)SL_ZH * P + p] = (float)saved[sv_at(ps, SV_ZH_G)];
}

// ... and back: the rows a SelfNorm-only backward reads (cnsn_mid_kernels.h::mid_bwd_a_kernel, load_cn_rows' constants for the
// rest) + the plane-order float rows of the channels-last tensor passes ([0] float(mean), [1..3] a_in = g, xr = 0, b_in = 0)
__global__ __launch_bounds__(kBlock) void nhwc_saved_from_slim_kernel(const float* __restrict__ slim, int N, int C, int relu,
                                                                      double* __restrict__ saved, float* __restrict__ rows) {
    const size_t P = (size_t)N * C, p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p < (size_t)C) {
        saved[(size_t)SV_ROWS * P + p] = slim_rstd(slim, P)[p];
        saved[(size_t)SV_ROWS * P + C + p] = 1.0;
    }
    if (p >= P) return;
    const SvRec ps = sv_rec_of_plane(p, N, C);
    const float hi = slim[(size_t)SL_MU_HI * P + p];
    const double mu = (double)hi + (double)slim[(size_t)SL_MU_LO * P + p];
    const float gate = slim[(size_t)SL_G * P + p];
    saved[sv_at(ps, SV_MU_C)] = mu;
    saved[sv_at(ps, SV_MU_P)] = mu;
    saved[sv_at(ps, SV_SIG_P)] = (double)slim[(size_t)SL_SIG * P + p];
    saved[sv_at(ps, SV_G)] = (double)gate;
    saved[sv_at(ps, SV_ZH_G)] = (double)slim[(size_t)SL_ZH * P + p];
    saved[sv_at(ps, SV_F)] = 1.0;
    saved[sv_at(ps, SV_ZH_F)] = 0.0;
    rows[p] = hi;
    if (relu) {
        rows[P + p] = gate;
        rows[2 * P + p] = 0.f;
        rows[3 * P + p] = 0.f;
    }
}

}  // namespace cnsn
