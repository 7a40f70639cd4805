// Channels-last ("NHWC") two-pass kernels (round 5): the tensor passes of the op for activations whose memory order is
// [n][h][w][c] — what MIOpen's NHWC convolutions produce and consume (torch.channels_last).
//
// Why: ResNet-50 bs 256 bf16 WITHOUT any CNSN unit runs 4 085 img/s in NCHW and 5 938 img/s in channels-last on this stack
// (tools/nhwc_probe.py, profiles/r05_nhwc.md): the NCHW-only op pinned the whole network to the slower convolutions.  The
// reference takes any layout through `.contiguous()` (models/cnsn.py:14,16); here a channels-last call is computed where it
// lies — no transpose — by these kernels around the SAME mid kernels (per-plane scalars: cnsn_mid_kernels.h), so the algebra,
// the `saved` contract and the parameter gradients are the two-pass strategy's.
//
// Layout: element (n, c, p) — p the pixel h*W + w — lives at ((n*M + p)*C + c).  A plane's statistics are a COLUMN reduction.
// A workgroup takes one instance n, one chunk of its pixels and up to 256 vector columns (a vector = 16 bytes = VEC adjacent
// channels of one pixel): thread (r, col) walks the pixels p0 + r, p0 + r + rows, ... of its column with full-width loads and
// keeps VEC accumulator pairs; the rows of a block are added in LDS in a fixed order, the pixel chunks of an instance by a
// finishing kernel in a fixed order: no atomics, results are reproducible.  Sums are taken about the plane's first pixel
// (the shift K of cnsn_stream_kernels.h), so S2 - S1^2/M does not cancel for planes with |mean| >> std.
// Un-boxed calls only (crop boxes would need the pixel's row / column per element): the host declines the others.
#pragma once
#include "cnsn_fused_stream_kernels.h"

#ifndef CNSN_NHWC_UB
#define CNSN_NHWC_UB 2  // pixels of a thread's column in flight at once in the backward passes (A/B: profiles/r05_nhwc.md)
#endif

namespace cnsn {

struct NhwcGeom {
    int N, C, M;
    int tc;      // vector columns of the tensor: C / VEC
    int tcb;     // vector columns a workgroup takes (<= 256)
    int rows;    // pixel rows a workgroup walks in parallel: 256 / tcb
    int ncb;     // column blocks: ceil(tc / tcb)
    int S;       // pixel chunks per instance
    int mchunk;  // pixels per chunk (the last one may be shorter)
    size_t P;    // planes: N * C
};

template <int VEC>
struct NhwcThread {
    int n, s, vc, r, p0, p1;
    bool active;
    __device__ __forceinline__ explicit NhwcThread(const NhwcGeom& g) : NhwcThread(g, (int)blockIdx.x) {}
    // tile number b = (n * S + s) * ncb + cb (the two-pass kernels launch one workgroup per tile; the single-launch kernels loop)
    __device__ __forceinline__ NhwcThread(const NhwcGeom& g, int b) {
        const int cb = b % g.ncb;
        b /= g.ncb;
        s = b % g.S;
        n = b / g.S;
        r = (int)threadIdx.x / g.tcb;
        vc = cb * g.tcb + (int)threadIdx.x % g.tcb;
        active = r < g.rows && vc < g.tc;
        p0 = s * g.mchunk;
        p1 = p0 + g.mchunk < g.M ? p0 + g.mchunk : g.M;
    }
    __device__ __forceinline__ size_t plane0(const NhwcGeom& g) const { return (size_t)n * g.C + (size_t)vc * VEC; }  // plane of channel 0 of the vector
    __device__ __forceinline__ size_t elem(const NhwcGeom& g, int p) const { return ((size_t)n * g.M + p) * g.C + (size_t)vc * VEC; }
};

// VEC adjacent per-plane floats (the planes of one pixel vector's channels) with 16-byte accesses.  Written as a loop over
// `p[j]` the backward passes compiled to one 4-byte load per channel and row — 56 loads a thread whose lanes lie 32 bytes apart,
// four times the cache-line requests of the tensor loads themselves (ISA of round 5's first version: profiles/r05_nhwc.md).
// p is 16-byte aligned: side arrays start on 256-byte boundaries and C is a whole number of vectors.
template <int VEC>
__device__ __forceinline__ void load_planes(const float* __restrict__ p, float (&o)[VEC]) {
    static_assert(VEC % 4 == 0, "whole float4s");
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        o[4 * q] = v.x, o[4 * q + 1] = v.y, o[4 * q + 2] = v.z, o[4 * q + 3] = v.w;
    }
}
template <int VEC>
__device__ __forceinline__ void store_planes(float* __restrict__ p, const float (&o)[VEC]) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// The same, AGENT-COHERENT: what one workgroup writes and another reads INSIDE a launch (the single-launch kernels of
// cnsn_nhwc_fused_kernels.h, across their grid barriers).  The L2s of the eight XCDs are not coherent with each other and a
// compute unit's L1 is never refreshed by another one's stores: these accesses carry `sc1` — stores written through to
// memory, loads served past the L1 — 16 bytes at a time through a buffer descriptor over the whole side array (aux bit 4 of
// the raw buffer builtins; a relaxed agent-scope atomic is the same instruction but at most 8 bytes wide, and an 8-byte
// write-through store is a fabric transaction of its own: PMC write traffic of the 7x7 sites 1.38-1.62 x the tensor's with
// those).  A full release / acquire fence pair in every wave instead cost these launches ~200 us per barrier.
struct CohBuf {
    __amdgpu_buffer_rsrc_t r;
    // `base`: wave-uniform (a kernel argument), 16-byte aligned; every index below is a multiple of 4 floats, < 2^29
    __device__ __forceinline__ explicit CohBuf(const float* base)
        : r(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0, 0x00020000)) {}
    __device__ __forceinline__ void st4(size_t idx, float a, float b, float c, float d) const {
        nt_u4 v;
        v.x = __float_as_uint(a), v.y = __float_as_uint(b), v.z = __float_as_uint(c), v.w = __float_as_uint(d);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(idx * 4), 0, 16);
    }
    __device__ __forceinline__ void ld4(size_t idx, float& a, float& b, float& c, float& d) const {
        const nt_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 4), 0, 16);
        a = __uint_as_float(v.x), b = __uint_as_float(v.y), c = __uint_as_float(v.z), d = __uint_as_float(v.w);
    }
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, float (&o)[VEC]) const {
        static_assert(VEC % 4 == 0, "whole float4s");
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) ld4(idx + 4 * q, o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
    template <int VEC>
    __device__ __forceinline__ void store(size_t idx, const float (&o)[VEC]) const {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) st4(idx + 4 * q, o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
};

// rows of a block -> one value per (column, channel-in-vector, accumulator), fixed order; lds: [NACC][rows][tcb*VEC] floats
// (COH: the partial sums are read by other workgroups of the same launch)
template <int VEC, int NACC, bool COH = false>
__device__ __forceinline__ void nhwc_rows_sum(const NhwcGeom& g, const NhwcThread<VEC>& t, float (&acc)[NACC][VEC], float* lds,
                                              float* __restrict__ part) {
    const int col = (int)threadIdx.x % g.tcb, width = g.tcb * VEC;
    if (t.r < g.rows) {
#pragma unroll
        for (int k = 0; k < NACC; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j) lds[((size_t)k * g.rows + t.r) * width + col * VEC + j] = t.active ? acc[k][j] : 0.f;
    }
    __syncthreads();
    if constexpr (COH) {
        // every thread adds the rows of FOUR adjacent results and stores them with one 16-byte write-through store: a wave
        // writes 1 KB of contiguous partial sums (the rows in a fixed order, as below)
        const CohBuf pb(part);
        const int total = NACC * width, first = (t.vc - col) * VEC;  // channel of the tile's first column
        for (int ch = (int)threadIdx.x * 4; ch < total; ch += kBlock * 4) {
            const int k = ch / width, off = ch - k * width;
            if (first + off >= g.C) continue;  // (the last column block may be partly outside the tensor; C % 4 == 0)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < g.rows; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(lds + ((size_t)k * g.rows + q) * width + off);
                v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
            }
            pb.st4(((size_t)t.s * NACC + k) * g.P + (size_t)t.n * g.C + first + off, v.x, v.y, v.z, v.w);
        }
        return;
    }
    if (t.r == 0 && t.active) {
        const size_t p = t.plane0(g);
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            float v[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                v[j] = 0.f;
                for (int q = 0; q < g.rows; ++q) v[j] += lds[((size_t)k * g.rows + q) * width + col * VEC + j];
            }
            store_planes<VEC>(part + ((size_t)t.s * NACC + k) * g.P + p, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass A: per-(n, c) sums of (X - K), (X - K)^2 over a pixel chunk; X = x [+ addend] (ADD_PRE), K = X at pixel 0
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int ADD>
__global__ __launch_bounds__(kBlock) void nhwc_stats_kernel(const T* __restrict__ x, const T* __restrict__ addend, NhwcGeom g,
                                                            float* __restrict__ part, float* __restrict__ kshift,
                                                            T* __restrict__ sum_out) {
    extern __shared__ float lds[];
    const NhwcThread<VEC> t(g);
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
        int p = t.p0 + t.r;
        for (; p + (U - 1) * g.rows < t.p1; p += U * g.rows) {
            Vec<T, VEC> va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t e = t.elem(g, p + u * g.rows);
                va[u] = load_vec_nt<T, VEC>(x + e);
                if constexpr (ADD == ADD_PRE) vb[u] = load_vec_nt<T, VEC>(addend + e);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                Vec<T, VEC> keep;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float X = ADD == ADD_PRE ? sum_t<T>(to_float(va[u].v[j]), to_float(vb[u].v[j])) : to_float(va[u].v[j]);
                    keep.v[j] = from_float<T>(X);  // (exact: sum_t rounds to T)
                    const float d = X - K[j];
                    acc[0][j] += d;
                    acc[1][j] = fmaf(d, d, acc[1][j]);
                }
                if constexpr (ADD == ADD_PRE) {
                    if (sum_out) store_vec<T, VEC>(sum_out + t.elem(g, p + u * g.rows), keep);  // (default policy: the apply pass reads it next)
                }
            }
        }
        for (; p < t.p1; p += g.rows) {
            const size_t e = t.elem(g, p);
            const Vec<T, VEC> va = load_vec_nt<T, VEC>(x + e);
            Vec<T, VEC> vb = va;
            if constexpr (ADD == ADD_PRE) vb = load_vec_nt<T, VEC>(addend + e);
            Vec<T, VEC> keep;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float X = ADD == ADD_PRE ? sum_t<T>(to_float(va.v[j]), to_float(vb.v[j])) : to_float(va.v[j]);
                keep.v[j] = from_float<T>(X);
                const float d = X - K[j];
                acc[0][j] += d;
                acc[1][j] = fmaf(d, d, acc[1][j]);
            }
            if constexpr (ADD == ADD_PRE) {
                if (sum_out) store_vec<T, VEC>(sum_out + e, keep);
            }
        }
        if (t.s == 0 && t.r == 0) {
            const size_t pl = t.plane0(g);
            store_planes<VEC>(kshift + pl, K);
        }
    }
    nhwc_rows_sum<VEC, 2>(g, t, acc, lds, part);
}

// the pixel chunks of a plane, in order: moments (double) for the mid kernel
static __global__ __launch_bounds__(kBlock) void nhwc_finish_stats_kernel(const float* __restrict__ part, const float* __restrict__ kshift,
                                                                   int S, size_t P, int M, double* __restrict__ mom) {
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
        s1 += (double)part[((size_t)s * 2 + 0) * P + p];
        s2 += (double)part[((size_t)s * 2 + 1) * P + p];
    }
    const double t = s2 - s1 * s1 / (double)M;
    mom[p] = (double)kshift[p] + s1 / (double)M;
    mom[P + p] = t > 0.0 ? t : 0.0;
}

// ... and the two backward sums (float rows as bwd_reduce_kernel writes them)
static __global__ __launch_bounds__(kBlock) void nhwc_finish_sums_kernel(const float* __restrict__ part, int S, size_t P,
                                                                  float* __restrict__ sums) {
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
        s1 += (double)part[((size_t)s * 2 + 0) * P + p];
        s2 += (double)part[((size_t)s * 2 + 1) * P + p];
    }
    sums[p] = (float)s1;
    sums[P + p] = (float)s2;
}

// ------------------------------------------------------------------------------------------------
// pass B: y = act(a_in * (X - xr) + b_in [+ addend]), coefficients per plane (FC rows of the mid kernel)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int ADD>
__global__ __launch_bounds__(kBlock) void nhwc_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ addend,
                                                                T* __restrict__ y, NhwcGeom g, ApplyCoef cf, int relu) {
    const NhwcThread<VEC> t(g);
    if (!t.active) return;
    const size_t pl = t.plane0(g);
    float a_in[VEC], xr[VEC], b_in[VEC];
    load_planes<VEC>(cf.a_in + pl, a_in);
    load_planes<VEC>(cf.xr + pl, xr);
    load_planes<VEC>(cf.b_in + pl, b_in);
    constexpr int U = ADD == ADD_NONE ? 4 : 2;
    auto emit = [&](const Vec<T, VEC>& va, const Vec<T, VEC>& vb, size_t e) {
        Vec<T, VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float f = to_float(va.v[j]);
            if constexpr (ADD == ADD_PRE) f = sum_t<T>(f, to_float(vb.v[j]));
            float v = fmaf(a_in[j], f - xr[j], b_in[j]);
            if constexpr (ADD == ADD_POST) v += to_float(vb.v[j]);
            o.v[j] = from_float<T>(relu ? fmaxf(v, 0.f) : v);
        }
        store_vec_nt<T, VEC>(y + e, o);
    };
    int p = t.p0 + t.r;
    for (; p + (U - 1) * g.rows < t.p1; p += U * g.rows) {
        Vec<T, VEC> va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = t.elem(g, p + u * g.rows);
            va[u] = load_vec_nt<T, VEC>(x + e);
            if constexpr (ADD != ADD_NONE) vb[u] = load_vec_nt<T, VEC>(addend + e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) emit(va[u], ADD != ADD_NONE ? vb[u] : va[u], t.elem(g, p + u * g.rows));
    }
    for (; p < t.p1; p += g.rows) {
        const size_t e = t.elem(g, p);
        const Vec<T, VEC> va = load_vec_nt<T, VEC>(x + e);
        Vec<T, VEC> vb = va;
        if constexpr (ADD != ADD_NONE) vb = load_vec_nt<T, VEC>(addend + e);
        emit(va, vb, e);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, first: the `saved` rows the stream kernels need, in plane order (p = n*C + c) as floats — `saved` is stored channel by
// channel and row by row over the batch (cnsn_layout.h), which a thread that owns VEC adjacent CHANNELS would read 8 bytes at a
// time from 64-byte sectors in every pixel chunk; once here instead.  rows: [0] float(mu_c), [1..3] a_in, xr, b_in (ReLU only)
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kBlock) void nhwc_saved_rows_kernel(const double* __restrict__ saved, int N, int C, int relu,
                                                                 float* __restrict__ rows) {
    const size_t P = (size_t)N * C, p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    const SvRec ps = sv_rec_of_plane(p, N, C);
    rows[p] = (float)saved[sv_at(ps, SV_MU_C)];
    if (relu) {
        rows[P + p] = (float)saved[sv_at(ps, SV_FC0 + FC_A_IN)];
        rows[2 * P + p] = (float)saved[sv_at(ps, SV_FC0 + FC_XR)];
        rows[3 * P + p] = (float)saved[sv_at(ps, SV_FC0 + FC_B_IN)];
    }
}

// the (masked) upstream gradient and the op's input of one element (un-boxed form of masked_pair)
template <typename T, int ADD>
__device__ __forceinline__ void nhwc_pair(float Gin, float xin, float bin, float a_in, float xr, float b_in, int relu, float& G,
                                          float& X) {
    X = ADD == ADD_PRE ? sum_t<T>(xin, bin) : xin;
    G = Gin;
    if (relu) {
        float t = fmaf(a_in, X - xr, b_in);
        if (ADD == ADD_POST) t += bin;
        G = relu_open<T>(t) ? Gin : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// pass A': per-(n, c) sums of G and G * (X - float(mu_c)) over a pixel chunk
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int ADD>
__global__ __launch_bounds__(kBlock) void nhwc_bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                 const T* __restrict__ addend, NhwcGeom g,
                                                                 const float* __restrict__ rows, int relu, float* __restrict__ part) {
    extern __shared__ float lds[];
    const NhwcThread<VEC> t(g);
    float acc[2][VEC], si[VEC], a_in[VEC], xr[VEC], b_in[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = si[j] = a_in[j] = xr[j] = b_in[j] = 0.f;
    if (t.active) {
        const size_t pl = t.plane0(g);
        load_planes<VEC>(rows + pl, si);
        if (relu) {
            load_planes<VEC>(rows + g.P + pl, a_in);
            load_planes<VEC>(rows + 2 * g.P + pl, xr);
            load_planes<VEC>(rows + 3 * g.P + pl, b_in);
        }
        constexpr int U = CNSN_NHWC_UB;
        auto eat = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, const Vec<T, VEC>& vb) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float G, X;
                nhwc_pair<T, ADD>(to_float(vg.v[j]), to_float(vx.v[j]), to_float(vb.v[j]), a_in[j], xr[j], b_in[j], relu, G, X);
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
                vg[u] = load_vec_nt<T, VEC>(gy + e);
                vx[u] = load_vec_nt<T, VEC>(x + e);
                if constexpr (ADD != ADD_NONE) vb[u] = load_vec_nt<T, VEC>(addend + e);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) eat(vg[u], vx[u], ADD != ADD_NONE ? vb[u] : vx[u]);
        }
        for (; p < t.p1; p += g.rows) {
            const size_t e = t.elem(g, p);
            const Vec<T, VEC> vg = load_vec_nt<T, VEC>(gy + e), vx = load_vec_nt<T, VEC>(x + e);
            Vec<T, VEC> vb = vx;
            if constexpr (ADD != ADD_NONE) vb = load_vec_nt<T, VEC>(addend + e);
            eat(vg, vx, vb);
        }
    }
    nhwc_rows_sum<VEC, 2>(g, t, acc, lds, part);
}

// ------------------------------------------------------------------------------------------------
// pass B': dx = cG * G + cX * (X - xr) + c0 (rows 0..3 of the backward coefficient block); ADD_POST + ReLU also writes the masked
// gradient (= gradient of the addend)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int ADD>
__global__ __launch_bounds__(kBlock) void nhwc_apply_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                const T* __restrict__ addend, T* __restrict__ dx,
                                                                T* __restrict__ d_addend, NhwcGeom g, const float* __restrict__ coef,
                                                                const float* __restrict__ rows, int relu) {
    const NhwcThread<VEC> t(g);
    if (!t.active) return;
    const size_t pl = t.plane0(g), P = g.P;
    float cG[VEC], cX[VEC], xri[VEC], c0[VEC], a_in[VEC], xr[VEC], b_in[VEC];
    load_planes<VEC>(coef + pl, cG);
    load_planes<VEC>(coef + P + pl, cX);
    load_planes<VEC>(coef + 2 * P + pl, xri);
    load_planes<VEC>(coef + 3 * P + pl, c0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a_in[j] = xr[j] = b_in[j] = 0.f;
    if (relu) {
        load_planes<VEC>(rows + P + pl, a_in);
        load_planes<VEC>(rows + 2 * P + pl, xr);
        load_planes<VEC>(rows + 3 * P + pl, b_in);
    }
    auto emit = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, const Vec<T, VEC>& vb, size_t e) {
        Vec<T, VEC> o, om;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float G, X;
            nhwc_pair<T, ADD>(to_float(vg.v[j]), to_float(vx.v[j]), to_float(vb.v[j]), a_in[j], xr[j], b_in[j], relu, G, X);
            o.v[j] = from_float<T>(fmaf(cG[j], G, fmaf(cX[j], X - xri[j], c0[j])));
            om.v[j] = from_float<T>(G);
        }
        store_vec_nt<T, VEC>(dx + e, o);
        if constexpr (ADD == ADD_POST) {
            if (d_addend) store_vec_nt<T, VEC>(d_addend + e, om);
        }
    };
    constexpr int U = CNSN_NHWC_UB;
    int p = t.p0 + t.r;
    for (; p + (U - 1) * g.rows < t.p1; p += U * g.rows) {
        Vec<T, VEC> vg[U], vx[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = t.elem(g, p + u * g.rows);
            vg[u] = load_vec_nt<T, VEC>(gy + e);
            vx[u] = load_vec_nt<T, VEC>(x + e);
            if constexpr (ADD != ADD_NONE) vb[u] = load_vec_nt<T, VEC>(addend + e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) emit(vg[u], vx[u], ADD != ADD_NONE ? vb[u] : vx[u], t.elem(g, p + u * g.rows));
    }
    for (; p < t.p1; p += g.rows) {
        const size_t e = t.elem(g, p);
        const Vec<T, VEC> vg = load_vec_nt<T, VEC>(gy + e), vx = load_vec_nt<T, VEC>(x + e);
        Vec<T, VEC> vb = vx;
        if constexpr (ADD != ADD_NONE) vb = load_vec_nt<T, VEC>(addend + e);
        emit(vg, vx, vb, e);
    }
}

}  // namespace cnsn
