// Streaming ("two-pass") kernels: every tensor pass of the fused op as one bandwidth-bound launch.
//
//   forward : plane_stats_kernel (read x)  -> mid_fwd_kernel (N*C scalars) -> apply_fwd_kernel (read x, write y)
//   backward: bwd_reduce_kernel (read G,x) -> mid_bwd_a/b (N*C scalars)    -> apply_bwd_kernel (read G,x, write dx)
//
// Thread mapping (all kernels): a plane (H*W contiguous elements of one (n,c)) is owned by LPP
// consecutive lanes — 16 (one DPP row), 64 (one wave) or 256 (the block) — chosen on the host from
// the plane size so that every lane issues full-width vector loads (VEC elements = up to 16 B) and
// the per-plane reduction never leaves registers/DPP for LPP <= 64.  Loads are issued UNROLL at a
// time before any is consumed, so each lane keeps UNROLL*16 B in flight.
#pragma once
#include "cnsn_device.h"



namespace cnsn {

#ifndef CNSN_UNROLL
#define CNSN_UNROLL 4
#endif
constexpr int kUnroll = CNSN_UNROLL;

// stream one plane: consume(vec, vec_index)
template <typename T, int VEC, int LPP, bool NT = false, typename Consume>
__device__ __forceinline__ void stream1(const T* __restrict__ base, int nvec, int lane, Consume&& consume, bool keep = false) {
    // keep (workgroup-uniform): default cache policy although NT — the first pass over a tensor small enough for the
    // second pass to find it in L2 / the Infinity Cache
    if (NT && keep) {
        stream1<T, VEC, LPP, false>(base, nvec, lane, consume);
        return;
    }
    int i = lane;
    for (; i + (kUnroll - 1) * LPP < nvec; i += kUnroll * LPP) {
        Vec<T, VEC> v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
            v[u] = NT ? load_vec_nt<T, VEC>(base + (size_t)(i + u * LPP) * VEC) : load_vec<T, VEC>(base + (size_t)(i + u * LPP) * VEC);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) consume(v[u], i + u * LPP);
    }
    for (; i < nvec; i += LPP) {
        Vec<T, VEC> v = NT ? load_vec_nt<T, VEC>(base + (size_t)i * VEC) : load_vec<T, VEC>(base + (size_t)i * VEC);
        consume(v, i);
    }
}

// stream two planes in lock step: consume(vecA, vecB, vec_index)
template <typename T, int VEC, int LPP, bool NT = false, typename Consume>
__device__ __forceinline__ void stream2(const T* __restrict__ a, const T* __restrict__ b, int nvec, int lane,
                                        Consume&& consume, bool keep = false) {
    if (NT && keep) {
        stream2<T, VEC, LPP, false>(a, b, nvec, lane, consume);
        return;
    }
    constexpr int U = kUnroll / 2;
    int i = lane;
    for (; i + (U - 1) * LPP < nvec; i += U * LPP) {
        Vec<T, VEC> va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            va[u] = NT ? load_vec_nt<T, VEC>(a + (size_t)(i + u * LPP) * VEC) : load_vec<T, VEC>(a + (size_t)(i + u * LPP) * VEC);
            vb[u] = NT ? load_vec_nt<T, VEC>(b + (size_t)(i + u * LPP) * VEC) : load_vec<T, VEC>(b + (size_t)(i + u * LPP) * VEC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) consume(va[u], vb[u], i + u * LPP);
    }
    for (; i < nvec; i += LPP) {
        Vec<T, VEC> va = NT ? load_vec_nt<T, VEC>(a + (size_t)i * VEC) : load_vec<T, VEC>(a + (size_t)i * VEC);
        Vec<T, VEC> vb = NT ? load_vec_nt<T, VEC>(b + (size_t)i * VEC) : load_vec<T, VEC>(b + (size_t)i * VEC);
        consume(va, vb, i);
    }
}

typedef float pk2f __attribute__((ext_vector_type(2)));

template <int LPP>
struct PlaneId {
    int p;       // plane index (clamped so that idle lanes still take part in reductions)
    int lane;    // lane within the plane's group
    bool valid;  // p is a real plane
    __device__ __forceinline__ explicit PlaneId(int P) {
        constexpr int PPB = kBlock / LPP;
        const int q = blockIdx.x * PPB + threadIdx.x / LPP;
        lane = threadIdx.x % LPP;
        valid = q < P;
        p = valid ? q : P - 1;
    }
};

// ------------------------------------------------------------------------------------------------
// pass A: plane statistics (reference: calc_ins_mean_std, models/cnsn.py:14-16)
//   un-boxed: out[0]=mean, out[1]=M2 of the whole plane
//   boxed   : out[0..5] = mean/M2 inside the content box, outside it, inside the style box
//   fin != NULL: fin[0]=mean, fin[1]=sqrt(M2/(cnt-1)+eps) of the (content-box) region, float32
//   moments are written as double: the mid kernels keep every per-plane scalar in double
// Sums are taken about a per-plane shift K (mean of the plane's first VEC elements) so that
// S2 - S1^2/cnt does not cancel for planes with |mean| >> std (post-ReLU activations).
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED>
__global__ __launch_bounds__(kBlock) void plane_stats_kernel(const T* __restrict__ x, Geom g,
                                                             double* __restrict__ mom, float* __restrict__ fin,
                                                             float eps) {
    constexpr int NACC = BOXED ? 6 : 2;
    __shared__ float lds[4 * NACC];
    const PlaneId<LPP> id(g.P);
    const T* base = x + (size_t)id.p * g.M;

    float K = 0.f;
    {
        const Vec<T, VEC> f = load_vec<T, VEC>(base);
#pragma unroll
        for (int j = 0; j < VEC; ++j) K += to_float(f.v[j]);
        K *= (1.0f / VEC);
    }
    // one accumulator per vector slot: VEC short chains instead of one long one (rounding error of
    // a sequential fp32 sum grows with its length), folded pairwise afterwards
    float part[NACC][VEC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) part[k][j] = 0.f;

    stream1<T, VEC, LPP, true>(base, g.nvec, id.lane, [&](const Vec<T, VEC>& v, int i) {
        if constexpr (!BOXED) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float d = to_float(v.v[j]) - K;
                part[0][j] += d;
                part[1][j] = fmaf(d, d, part[1][j]);
            }
        } else {
            // boxed: the kernel is VALU-bound for 16-bit tensors, so the per-element work is kept minimal — the row
            // tests are per vector (VEC divides the width: a vector lies in one row), membership becomes a 0/1
            // factor, and the sums OUTSIDE the content box are obtained as (whole plane) - (inside) afterwards
            // (all sums are about the same shift K, so the subtraction does not cancel catastrophically; what it
            // loses is scaled by the weight Mo/M the outside region has in the merged statistics)
            const int e = i * VEC;
            const int r = e / g.Wd, c = e - r * g.Wd;
            const bool row_c = (unsigned)(r - g.cb.r0) < (unsigned)(g.cb.r1 - g.cb.r0);
            const bool row_s = (unsigned)(r - g.sb.r0) < (unsigned)(g.sb.r1 - g.sb.r0);
            const unsigned wc = row_c ? (unsigned)(g.cb.c1 - g.cb.c0) : 0u, ws = row_s ? (unsigned)(g.sb.c1 - g.sb.c0) : 0u;
            const int cc = c - g.cb.c0, cs = c - g.sb.c0;
            if constexpr (VEC >= 2) {
                // two elements per instruction (v_pk_add/mul/fma_f32): the accumulators of slots j, j+1 as a pair
#pragma unroll
                for (int j = 0; j < VEC; j += 2) {
                    const pk2f d = pk2f{to_float(v.v[j]), to_float(v.v[j + 1])} - pk2f{K, K};
                    const pk2f d2 = d * d;
                    const pk2f mc = {(unsigned)(cc + j) < wc ? 1.f : 0.f, (unsigned)(cc + j + 1) < wc ? 1.f : 0.f};
                    const pk2f ms = {(unsigned)(cs + j) < ws ? 1.f : 0.f, (unsigned)(cs + j + 1) < ws ? 1.f : 0.f};
                    pk2f a0 = {part[0][j], part[0][j + 1]}, a1 = {part[1][j], part[1][j + 1]};
                    pk2f a2 = {part[2][j], part[2][j + 1]}, a3 = {part[3][j], part[3][j + 1]};
                    pk2f a4 = {part[4][j], part[4][j + 1]}, a5 = {part[5][j], part[5][j + 1]};
                    a0 = __builtin_elementwise_fma(mc, d, a0);
                    a1 = __builtin_elementwise_fma(mc, d2, a1);
                    a2 += d;   // whole plane here; the inside is subtracted below
                    a3 += d2;
                    a4 = __builtin_elementwise_fma(ms, d, a4);
                    a5 = __builtin_elementwise_fma(ms, d2, a5);
                    part[0][j] = a0.x, part[0][j + 1] = a0.y, part[1][j] = a1.x, part[1][j + 1] = a1.y;
                    part[2][j] = a2.x, part[2][j + 1] = a2.y, part[3][j] = a3.x, part[3][j + 1] = a3.y;
                    part[4][j] = a4.x, part[4][j + 1] = a4.y, part[5][j] = a5.x, part[5][j + 1] = a5.y;
                }
            } else {
                const float d = to_float(v.v[0]) - K;
                const float d2 = d * d;
                const float mc = (unsigned)cc < wc ? 1.f : 0.f, ms = (unsigned)cs < ws ? 1.f : 0.f;
                part[0][0] = fmaf(mc, d, part[0][0]);
                part[1][0] = fmaf(mc, d2, part[1][0]);
                part[2][0] += d;
                part[3][0] += d2;
                part[4][0] = fmaf(ms, d, part[4][0]);
                part[5][0] = fmaf(ms, d2, part[5][0]);
            }
        }
    }, g.keep != 0);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
#pragma unroll
        for (int w = VEC / 2; w > 0; w >>= 1)
#pragma unroll
            for (int j = 0; j < w; ++j) part[k][j] += part[k][j + w];
        acc[k] = part[k][0];
    }
    group_sum<LPP, NACC>(acc, lds);
    if constexpr (BOXED) {  // outside = whole plane - inside
        acc[2] -= acc[0];
        acc[3] -= acc[1];
    }

    if (id.lane == 0 && id.valid) {
        auto moments = [&](float s1, float s2, int cnt, double& mean, double& m2) {
            if (cnt <= 0) {
                mean = 0.0;
                m2 = 0.0;
                return;
            }
            const double d1 = s1, d2 = s2;
            mean = double(K) + d1 / cnt;
            const double t = d2 - d1 * d1 / cnt;
            m2 = t > 0.0 ? t : 0.0;
        };
        const size_t P = g.P;
        const int Mc = BOXED ? g.cb.area() : g.M;
        double mean, m2;
        moments(acc[0], acc[1], Mc, mean, m2);
        if (fin) {  // stand-alone calc_ins_mean_std: (mean, std) of the (boxed) region as float32
            fin[id.p] = (float)mean;
            fin[P + id.p] = (float)sqrt(m2 / double(Mc - 1) + (double)eps);
        } else {
            mom[id.p] = mean;
            mom[P + id.p] = m2;
            if constexpr (BOXED) {
                moments(acc[2], acc[3], g.M - Mc, mean, m2);
                mom[2 * P + id.p] = mean;
                mom[3 * P + id.p] = m2;
                moments(acc[4], acc[5], g.sb.area(), mean, m2);
                mom[4 * P + id.p] = mean;
                mom[5 * P + id.p] = m2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass B: y = a_in*(x - xr) + b_in inside the content box, a_out*x + b_out outside
// (reference: instance_norm_mix :27-29, the box paste :75-82, the lam blend :87, SelfNorm :148/:150
//  — all folded into five per-plane coefficients by mid_fwd_kernel).  xr may be NULL (= 0).
// ------------------------------------------------------------------------------------------------
struct ApplyCoef {
    const float* a_in;
    const float* xr;
    const float* b_in;
    const float* a_out;
    const float* b_out;
};

template <typename T, int VEC, int LPP, bool BOXED>
__global__ __launch_bounds__(kBlock) void apply_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, Geom g,
                                                           ApplyCoef cf) {
    const PlaneId<LPP> id(g.P);
    if (!id.valid) return;
    const size_t off = (size_t)id.p * g.M;
    const float a_in = cf.a_in[id.p], xr = cf.xr ? cf.xr[id.p] : 0.f, b_in = cf.b_in[id.p];
    float a_out = 0.f, b_out = 0.f;
    if constexpr (BOXED) {
        a_out = cf.a_out[id.p];
        b_out = cf.b_out[id.p];
    }
    T* yb = y + off;
    stream1<T, VEC, LPP, true>(x + off, g.nvec, id.lane, [&](const Vec<T, VEC>& v, int i) {
        Vec<T, VEC> o;
        if constexpr (!BOXED) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = from_float<T>(fmaf(a_in, to_float(v.v[j]) - xr, b_in));
        } else {
            const int e = i * VEC;
            const int r = e / g.Wd, c = e - r * g.Wd;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float f = to_float(v.v[j]);
                o.v[j] = from_float<T>(g.cb.has(r, c + j) ? fmaf(a_in, f - xr, b_in) : fmaf(a_out, f, b_out));
            }
        }
        store_vec_nt<T, VEC>(yb + (size_t)i * VEC, o);
    });
}

// ------------------------------------------------------------------------------------------------
// pass A': per-plane sums of the upstream gradient G against x
//   un-boxed: out[0]=sum G, out[1]=sum G*(x-shift_in)
//   boxed   : out[0..3] = the same inside the content box (shift_in) and outside it (shift_out)
// shift pointers may be NULL (= 0).
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED>
__global__ __launch_bounds__(kBlock) void bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ x, Geom g,
                                                            const double* __restrict__ shift_in,
                                                            const double* __restrict__ shift_out, int shift_stride,
                                                            float* __restrict__ out) {
    constexpr int NACC = BOXED ? 4 : 2;
    __shared__ float lds[4 * NACC];
    const PlaneId<LPP> id(g.P);
    const size_t off = (size_t)id.p * g.M;
    // shifts: a plain array over the planes (shift_stride 1), or rows of `saved` (shift_stride 0: the pointers are the
    // rows' starts, cnsn_layout.h); mid_bwd_a undoes the rounding
    const size_t srec = shift_stride == 1 ? (size_t)id.p : sv_rec_of_plane((size_t)id.p, g.N, g.C).base;
    const float si = shift_in ? (float)shift_in[srec] : 0.f;
    const float so = (BOXED && shift_out) ? (float)shift_out[srec] : 0.f;
    float part[NACC][VEC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) part[k][j] = 0.f;
    stream2<T, VEC, LPP, true>(gy + off, x + off, g.nvec, id.lane,
                         [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, int i) {
                             if constexpr (!BOXED) {
#pragma unroll
                                 for (int j = 0; j < VEC; ++j) {
                                     const float G = to_float(vg.v[j]);
                                     part[0][j] += G;
                                     part[1][j] = fmaf(G, to_float(vx.v[j]) - si, part[1][j]);
                                 }
                             } else {  // (see plane_stats_kernel: row test per vector, 0/1 factor, outside = whole - inside)
                                 const int e = i * VEC;
                                 const int r = e / g.Wd, c = e - r * g.Wd;
                                 const unsigned wc = (unsigned)(r - g.cb.r0) < (unsigned)(g.cb.r1 - g.cb.r0)
                                                         ? (unsigned)(g.cb.c1 - g.cb.c0)
                                                         : 0u;
#pragma unroll
                                 for (int j = 0; j < VEC; ++j) {
                                     const float G = to_float(vg.v[j]), X = to_float(vx.v[j]);
                                     const float mc = (unsigned)(c + j - g.cb.c0) < wc ? 1.f : 0.f;
                                     const float pr = G * (X - si);
                                     part[0][j] = fmaf(mc, G, part[0][j]);
                                     part[1][j] = fmaf(mc, pr, part[1][j]);
                                     part[2][j] += G;
                                     part[3][j] += pr;
                                 }
                             }
                         }, g.keep != 0);
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
#pragma unroll
        for (int w = VEC / 2; w > 0; w >>= 1)
#pragma unroll
            for (int j = 0; j < w; ++j) part[k][j] += part[k][j + w];
        acc[k] = part[k][0];
    }
    group_sum<LPP, NACC>(acc, lds);
    if constexpr (BOXED) {  // sum_out G*(X-so) = sum_out G*(X-si) - (so-si)*sum_out G
        acc[2] -= acc[0];
        acc[3] = (acc[3] - acc[1]) - (so - si) * acc[2];
    }
    if (id.lane == 0 && id.valid) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) out[(size_t)k * g.P + id.p] = acc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// pass B': dx = cG*G + cX*(x - xr) + c0, coefficients by region (inside / outside the content
// box), plus eS*(x - xs) + e0 inside the style box (the gradient a plane receives for having been
// another instance's style source).  Coefficients: SoA rows of `coef`, stride P (see mid_bwd_b).
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED>
__global__ __launch_bounds__(kBlock) void apply_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                           T* __restrict__ dx, Geom g,
                                                           const float* __restrict__ coef) {
    const PlaneId<LPP> id(g.P);
    if (!id.valid) return;
    const size_t off = (size_t)id.p * g.M;
    const size_t P = g.P;
    const float cG_i = coef[id.p], cX_i = coef[P + id.p], xr_i = coef[2 * P + id.p], c0_i = coef[3 * P + id.p];
    float cG_o = 0.f, cX_o = 0.f, xr_o = 0.f, c0_o = 0.f, eS = 0.f, xs = 0.f, e0 = 0.f;
    if constexpr (BOXED) {
        cG_o = coef[4 * P + id.p];
        cX_o = coef[5 * P + id.p];
        xr_o = coef[6 * P + id.p];
        c0_o = coef[7 * P + id.p];
        eS = coef[8 * P + id.p];
        xs = coef[9 * P + id.p];
        e0 = coef[10 * P + id.p];
    }
    T* db = dx + off;
    stream2<T, VEC, LPP, true>(gy + off, x + off, g.nvec, id.lane,
                         [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, int i) {
                             Vec<T, VEC> o;
                             if constexpr (!BOXED) {
#pragma unroll
                                 for (int j = 0; j < VEC; ++j) {
                                     const float G = to_float(vg.v[j]), X = to_float(vx.v[j]);
                                     o.v[j] = from_float<T>(fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i)));
                                 }
                             } else {
                                 const int e = i * VEC;
                                 const int r = e / g.Wd, c = e - r * g.Wd;
#pragma unroll
                                 for (int j = 0; j < VEC; ++j) {
                                     const float G = to_float(vg.v[j]), X = to_float(vx.v[j]);
                                     const bool ic = g.cb.has(r, c + j), is = g.sb.has(r, c + j);
                                     float d = ic ? fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i))
                                                  : fmaf(cG_o, G, fmaf(cX_o, X - xr_o, c0_o));
                                     d += is ? fmaf(eS, X - xs, e0) : 0.f;
                                     o.v[j] = from_float<T>(d);
                                 }
                             }
                             store_vec_nt<T, VEC>(db + (size_t)i * VEC, o);
                         });
}

// ------------------------------------------------------------------------------------------------
// backward of calc_ins_mean_std alone: dx = dmean/cnt + dstd*(x-mean)/(std*(cnt-1)) in the box, else 0
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED>
__global__ __launch_bounds__(kBlock) void plane_stats_bwd_kernel(const T* __restrict__ x, T* __restrict__ dx, Geom g,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ std,
                                                                 const float* __restrict__ dmean,
                                                                 const float* __restrict__ dstd) {
    const PlaneId<LPP> id(g.P);
    if (!id.valid) return;
    const size_t off = (size_t)id.p * g.M;
    const int cnt = BOXED ? g.cb.area() : g.M;
    const float mu = mean[id.p];
    const float cX = dstd[id.p] / (std[id.p] * float(cnt - 1));
    const float c0 = dmean[id.p] / float(cnt);
    T* db = dx + off;
    stream1<T, VEC, LPP, true>(x + off, g.nvec, id.lane, [&](const Vec<T, VEC>& v, int i) {
        Vec<T, VEC> o;
        const int e = i * VEC;
        const int r = BOXED ? e / g.Wd : 0, c = BOXED ? e - r * g.Wd : 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float d = fmaf(cX, to_float(v.v[j]) - mu, c0);
            o.v[j] = from_float<T>((!BOXED || g.cb.has(r, c + j)) ? d : 0.f);
        }
        store_vec_nt<T, VEC>(db + (size_t)i * VEC, o);
    });
}

}  // namespace cnsn
