// Two-pass kernels for SMALL planes (7x7, 8x8, 14x14, ... : a plane of at most 1 KiB), "packed" access.
//
// The streaming kernels give every plane 16 or 64 lanes and let each lane load the widest vector that divides
// the plane.  For small planes that breaks down: a 7x7 bf16 plane is 98 bytes — not a multiple of anything, so
// the loads degrade to 2 bytes per lane — and a 14x14 plane keeps only 49 of 64 lanes busy with one load each
// (ResNet-50's last two stages, (N,1024,14,14) and (N,2048,7,7), ran at < 1 TB/s that way).
//
// Here a wave takes a RUN of R consecutive planes — one contiguous, 16-byte aligned piece of the tensor
// (R is chosen so that R*M*sizeof(T) is a multiple of 16 bytes) — and
//   1. copies it to LDS with full 16-byte vector loads (perfectly coalesced, several in flight per lane),
//   2. lets 16 lanes (one DPP row) work on one plane at a time out of LDS, element by element, whatever its
//      alignment; per-plane scalars (coefficients, shifts) for the whole run were fetched up front with the bulk
//      loads and sit in LDS too; per-plane sums are reduced inside the DPP row (deterministic),
//   3. (apply kernels) writes the results back into the same LDS positions and flushes the run to HBM with
//      16-byte vector stores.
// One set of kernels serves the op alone and the residual-block epilogue (ADD template flag, runtime relu).
// Side-array layouts (moments, sums, coefficient rows, `saved`) are those of the streaming kernels, so the mid
// kernels do not know the difference.
#pragma once
#include "cnsn_fused_stream_kernels.h"
#include "cnsn_layout.h"
#include "cnsn_packed.h"

namespace cnsn {

constexpr int kPackedWaves = kBlock / 64;

// bytes of dynamic LDS: per wave NT staged tensors + NSC per-plane scalars for R planes
__host__ __device__ inline size_t packed_lds_bytes(const PackedGeom& g, int nt, int nsc) {
    return (size_t)kPackedWaves * ((size_t)nt * g.run_vecs * 16 + (size_t)nsc * g.R * 4);
}

typedef unsigned pk_u4 __attribute__((ext_vector_type(4)));

// copy one run between HBM and LDS with 16-byte vectors; the (at most one) vector that crosses the end of the
// tensor is moved element by element, so nothing outside [0, total) is ever touched
template <typename T>
__device__ __forceinline__ void run_to_lds(const T* __restrict__ src, long long byte0, const PackedGeom& g, char* lds,
                                           int lane) {
    const char* s = (const char*)src + byte0;
    for (int v = lane; v < g.run_vecs; v += 64) {
        const long long end = byte0 + (long long)(v + 1) * 16;
        if (end <= g.total) {
            *(pk_u4*)(lds + v * 16) = __builtin_nontemporal_load((const pk_u4*)(s + v * 16));
        } else {
            const int left = (int)(g.total - (end - 16));  // bytes of this vector inside the tensor (may be <= 0)
            for (int b = 0; b < 16; b += (int)sizeof(T))
                if (b < left) *(T*)(lds + v * 16 + b) = *(const T*)(s + v * 16 + b);
        }
    }
}
template <typename T>
__device__ __forceinline__ void lds_to_run(T* __restrict__ dst, long long byte0, const PackedGeom& g, const char* lds,
                                           int lane) {
    char* d = (char*)dst + byte0;
    for (int v = lane; v < g.run_vecs; v += 64) {
        const long long end = byte0 + (long long)(v + 1) * 16;
        if (end <= g.total) {
            __builtin_nontemporal_store(*(const pk_u4*)(lds + v * 16), (pk_u4*)(d + v * 16));
        } else {
            const int left = (int)(g.total - (end - 16));
            for (int b = 0; b < 16; b += (int)sizeof(T))
                if (b < left) *(T*)(d + v * 16 + b) = *(const T*)(lds + v * 16 + b);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the driver.  Op provides:
//   static constexpr int NIN, NOUT, NSC;   staged input tensors, output tensors (out t reuses in t's LDS), scalars
//   void fetch(int p, float* sc) const;            per-plane scalars of plane p -> sc[0..NSC) (global loads)
//   Acc  begin(const float* sc, const float* first) const;   first[t] = element 0 of the plane in tensor t
//   void elem(Acc&, const float* sc, const float (&f)[NIN], bool ic, bool is, float (&o)[NOUT>0?NOUT:1]) const;
//   void end(Acc&, const float* sc, int p, bool leader) const;      after the 16-lane reduction inside
// ------------------------------------------------------------------------------------------------
template <typename T, bool BOXED, typename Op>
__global__ __launch_bounds__(kBlock) void packed_kernel(PackedGeom g, const T* __restrict__ in0,
                                                        const T* __restrict__ in1, const T* __restrict__ in2,
                                                        T* __restrict__ out0, T* __restrict__ out1, Op op) {
    constexpr int NIN = Op::NIN, NOUT = Op::NOUT, NSC = Op::NSC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, l16 = lane & 15;
    const int region = g.run_vecs * 16;
    char* stage = smem + (size_t)wave * ((size_t)NIN * region + (size_t)NSC * g.R * 4);
    float* scal = (float*)(stage + (size_t)NIN * region);
    const T* ins[3] = {in0, in1, in2};
    T* outs[2] = {out0, out1};

    for (int run0 = blockIdx.x * kPackedWaves; run0 < g.runs; run0 += gridDim.x * kPackedWaves) {
        const int run = run0 + wave;
        const bool active = run < g.runs;  // wave-uniform
        const long long byte0 = (long long)run * region;
        const int p0 = run * g.R;
        if (active) {
#pragma unroll
            for (int t = 0; t < NIN; ++t) run_to_lds<T>(ins[t], byte0, g, stage + (size_t)t * region, lane);
            if constexpr (NSC > 0) {
                if (lane < g.R && p0 + lane < g.P) op.fetch(p0 + lane, scal + lane * NSC);
            }
        }
        __syncthreads();
        if (active) {
            for (int pl = grp; pl < g.R; pl += 4) {
                const int p = p0 + pl;
                if (p >= g.P) break;  // uniform within the 16-lane group (and planes only grow with pl)
                const float* sc = scal + pl * NSC;
                const size_t eoff = (size_t)pl * g.M;
                float first[NIN];
#pragma unroll
                for (int t = 0; t < NIN; ++t) first[t] = to_float(((const T*)(stage + (size_t)t * region))[eoff]);
                auto acc = op.begin(sc, first);
                int r = 0, c = l16;  // (row, column) of this lane's element; only the boxed variants look at them
                if constexpr (BOXED) {
                    r = l16 / g.Wd;
                    c = l16 - r * g.Wd;
                }
                for (int e = l16; e < g.M; e += 16) {
                    float f[NIN];
#pragma unroll
                    for (int t = 0; t < NIN; ++t) f[t] = to_float(((const T*)(stage + (size_t)t * region))[eoff + e]);
                    const bool ic = !BOXED || g.cb.has(r, c), is = !BOXED || g.sb.has(r, c);
                    float o[NOUT > 0 ? NOUT : 1];
                    op.elem(acc, sc, f, ic, is, o);
#pragma unroll
                    for (int t = 0; t < NOUT; ++t) ((T*)(stage + (size_t)t * region))[eoff + e] = from_float<T>(o[t]);
                    if constexpr (BOXED) {
                        c += 16;
                        while (c >= g.Wd) {
                            c -= g.Wd;
                            ++r;
                        }
                    }
                }
                op.end(acc, sc, p, l16 == 0);
            }
        }
        if constexpr (NOUT > 0) {
            __syncthreads();
            if (active) {
#pragma unroll
                for (int t = 0; t < NOUT; ++t)
                    if (outs[t]) lds_to_run<T>(outs[t], byte0, g, stage + (size_t)t * region, lane);
            }
        }
        __syncthreads();  // the next iteration overwrites the staging area
    }
}

// ------------------------------------------------------------------------------------------------
// pass A: plane moments (layout of plane_stats_kernel's `mom`)
// ------------------------------------------------------------------------------------------------
template <typename T, bool BOXED, int ADD>
struct PackedStatsOp {
    static constexpr int NIN = ADD == ADD_PRE ? 2 : 1, NOUT = 0, NSC = 0;
    double* mom;
    int P, M, Mc, Ms;
    struct Acc {
        float K, a[BOXED ? 6 : 2];
    };
    __device__ __forceinline__ void fetch(int, float*) const {}
    __device__ __forceinline__ float x_of(const float (&f)[NIN]) const {
        if constexpr (ADD == ADD_PRE)
            return sum_t<T>(f[0], f[1]);
        else
            return f[0];
    }
    __device__ __forceinline__ Acc begin(const float*, const float* first) const {
        Acc a;
        float f[NIN];
#pragma unroll
        for (int t = 0; t < NIN; ++t) f[t] = first[t];
        a.K = x_of(f);  // shift: sums are taken about the plane's first element (no cancellation for |mean| >> std)
#pragma unroll
        for (int k = 0; k < (BOXED ? 6 : 2); ++k) a.a[k] = 0.f;
        return a;
    }
    __device__ __forceinline__ void elem(Acc& a, const float*, const float (&f)[NIN], bool ic, bool is, float (&)[1]) const {
        const float d = x_of(f) - a.K;
        if constexpr (!BOXED) {
            a.a[0] += d;
            a.a[1] = fmaf(d, d, a.a[1]);
        } else {
            const float d2 = d * d;
            a.a[0] += ic ? d : 0.f;
            a.a[1] += ic ? d2 : 0.f;
            a.a[2] += ic ? 0.f : d;
            a.a[3] += ic ? 0.f : d2;
            a.a[4] += is ? d : 0.f;
            a.a[5] += is ? d2 : 0.f;
        }
    }
    __device__ __forceinline__ void end(Acc& a, const float*, int p, bool leader) const {
#pragma unroll
        for (int k = 0; k < (BOXED ? 6 : 2); ++k) a.a[k] = row16_sum(a.a[k]);
        if (!leader) return;
        auto moments = [&](float s1, float s2, int cnt, double& mean, double& m2) {
            if (cnt <= 0) {
                mean = 0.0;
                m2 = 0.0;
                return;
            }
            const double d1 = s1, d2 = s2;
            mean = double(a.K) + d1 / cnt;
            const double t = d2 - d1 * d1 / cnt;
            m2 = t > 0.0 ? t : 0.0;
        };
        double mean, m2;
        moments(a.a[0], a.a[1], Mc, mean, m2);
        mom[p] = mean;
        mom[(size_t)P + p] = m2;
        if constexpr (BOXED) {
            moments(a.a[2], a.a[3], M - Mc, mean, m2);
            mom[2 * (size_t)P + p] = mean;
            mom[3 * (size_t)P + p] = m2;
            moments(a.a[4], a.a[5], Ms, mean, m2);
            mom[4 * (size_t)P + p] = mean;
            mom[5 * (size_t)P + p] = m2;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// pass B: y = act(affine(x [+ addend]) [+ addend]); coefficient rows of mid_fwd_kernel
// ------------------------------------------------------------------------------------------------
template <typename T, bool BOXED, int ADD>
struct PackedApplyFwdOp {
    static constexpr int NIN = ADD == ADD_NONE ? 1 : 2, NOUT = 1, NSC = 5;
    const float* coef;  // FC_ROWS rows of stride P
    int P, relu;
    struct Acc {};
    __device__ __forceinline__ void fetch(int p, float* sc) const {
#pragma unroll
        for (int r = 0; r < FC_ROWS; ++r) sc[r] = coef[(size_t)r * P + p];
    }
    __device__ __forceinline__ Acc begin(const float*, const float*) const { return Acc{}; }
    __device__ __forceinline__ void elem(Acc&, const float* sc, const float (&f)[NIN], bool ic, bool, float (&o)[1]) const {
        float x = f[0];
        if constexpr (ADD == ADD_PRE) x = sum_t<T>(x, f[1]);
        float t = ic ? fmaf(sc[FC_A_IN], x - sc[FC_XR], sc[FC_B_IN]) : fmaf(sc[FC_A_OUT], x, sc[FC_B_OUT]);
        if constexpr (ADD == ADD_POST) t += f[1];
        o[0] = relu ? fmaxf(t, 0.f) : t;
    }
    __device__ __forceinline__ void end(Acc&, const float*, int, bool) const {}
};

// scalars of the backward passes that come from `saved`
constexpr int kBwdScShift = 2;              // si, so
constexpr int kBwdScFwd = kBwdScShift + 5;  // + the forward's apply coefficients (ReLU mask)

template <typename T, bool BOXED, int ADD>
__device__ __forceinline__ void packed_masked_pair(const float* fc, int relu, float Gin, float xin, float bin, bool ic,
                                                   float& G, float& X) {
    X = ADD == ADD_PRE ? sum_t<T>(xin, bin) : xin;
    G = Gin;
    if (relu) {
        float t = ic ? fmaf(fc[FC_A_IN], X - fc[FC_XR], fc[FC_B_IN]) : fmaf(fc[FC_A_OUT], X, fc[FC_B_OUT]);
        if (ADD == ADD_POST) t += bin;
        G = relu_open<T>(t) ? Gin : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// pass A': sums of the (masked) gradient against x [+ addend]; rows of bwd_reduce_kernel
// ------------------------------------------------------------------------------------------------
template <typename T, bool BOXED, int ADD>
struct PackedReduceOp {
    static constexpr int NIN = ADD == ADD_NONE ? 2 : 3, NOUT = 0, NSC = kBwdScFwd;
    const double* saved;
    float* out;
    int P, relu, N, C;
    struct Acc {
        float a[BOXED ? 4 : 2];
    };
    __device__ __forceinline__ void fetch(int p, float* sc) const {
        const SvRec ps = sv_rec_of_plane((size_t)p, N, C);
        sc[0] = (float)saved[sv_at(ps, SV_MU_C)];
        sc[1] = BOXED ? (float)saved[sv_at(ps, SV_MU_O)] : 0.f;
        if (relu) {
#pragma unroll
            for (int r = 0; r < FC_ROWS; ++r) sc[kBwdScShift + r] = (float)saved[sv_at(ps, SV_FC0 + r)];
        }
    }
    __device__ __forceinline__ Acc begin(const float*, const float*) const {
        Acc a;
#pragma unroll
        for (int k = 0; k < (BOXED ? 4 : 2); ++k) a.a[k] = 0.f;
        return a;
    }
    __device__ __forceinline__ void elem(Acc& a, const float* sc, const float (&f)[NIN], bool ic, bool, float (&)[1]) const {
        float G, X;
        packed_masked_pair<T, BOXED, ADD>(sc + kBwdScShift, relu, f[0], f[1], NIN > 2 ? f[NIN - 1] : 0.f, ic, G, X);
        if constexpr (!BOXED) {
            a.a[0] += G;
            a.a[1] = fmaf(G, X - sc[0], a.a[1]);
        } else {
            a.a[0] += ic ? G : 0.f;
            a.a[1] += ic ? G * (X - sc[0]) : 0.f;
            a.a[2] += ic ? 0.f : G;
            a.a[3] += ic ? 0.f : G * (X - sc[1]);
        }
    }
    __device__ __forceinline__ void end(Acc& a, const float*, int p, bool leader) const {
#pragma unroll
        for (int k = 0; k < (BOXED ? 4 : 2); ++k) a.a[k] = row16_sum(a.a[k]);
        if (leader) {
#pragma unroll
            for (int k = 0; k < (BOXED ? 4 : 2); ++k) out[(size_t)k * P + p] = a.a[k];
        }
    }
};

// ------------------------------------------------------------------------------------------------
// pass B': dx (and, POST + ReLU, the masked gradient = gradient of the addend); rows of mid_bwd_b_kernel
// ------------------------------------------------------------------------------------------------
template <typename T, bool BOXED, int ADD>
struct PackedApplyBwdOp {
    static constexpr int NIN = ADD == ADD_NONE ? 2 : 3, NOUT = ADD == ADD_POST ? 2 : 1, NSC = BC_ROWS + 5;
    const float* coef;  // BC_ROWS rows of stride P
    const double* saved;
    int P, relu, N, C;
    struct Acc {};
    __device__ __forceinline__ void fetch(int p, float* sc) const {
#pragma unroll
        for (int r = 0; r < (BOXED ? (int)BC_ROWS : 4); ++r) sc[r] = coef[(size_t)r * P + p];
        if (relu) {
            const SvRec ps = sv_rec_of_plane((size_t)p, N, C);
#pragma unroll
            for (int r = 0; r < FC_ROWS; ++r) sc[BC_ROWS + r] = (float)saved[sv_at(ps, SV_FC0 + r)];
        }
    }
    __device__ __forceinline__ Acc begin(const float*, const float*) const { return Acc{}; }
    __device__ __forceinline__ void elem(Acc&, const float* sc, const float (&f)[NIN], bool ic, bool is,
                                         float (&o)[NOUT]) const {
        float G, X;
        packed_masked_pair<T, BOXED, ADD>(sc + BC_ROWS, relu, f[0], f[1], NIN > 2 ? f[NIN - 1] : 0.f, ic, G, X);
        float d;
        if constexpr (!BOXED) {
            d = fmaf(sc[BC_CG_IN], G, fmaf(sc[BC_CX_IN], X - sc[BC_XR_IN], sc[BC_C0_IN]));
        } else {
            d = ic ? fmaf(sc[BC_CG_IN], G, fmaf(sc[BC_CX_IN], X - sc[BC_XR_IN], sc[BC_C0_IN]))
                   : fmaf(sc[BC_CG_OUT], G, fmaf(sc[BC_CX_OUT], X - sc[BC_XR_OUT], sc[BC_C0_OUT]));
            d += is ? fmaf(sc[BC_ES], X - sc[BC_XS], sc[BC_E0]) : 0.f;
        }
        o[0] = d;
        if constexpr (NOUT > 1) o[1] = G;
    }
    __device__ __forceinline__ void end(Acc&, const float*, int, bool) const {}
};

}  // namespace cnsn
