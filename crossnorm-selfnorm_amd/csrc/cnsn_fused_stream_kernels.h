// Two-pass kernels with the residual-block epilogue folded in:
//     y = act( CNSN(x [+ addend]) [+ addend] ),   act = ReLU or identity
// (reference call sites: models/imagenet/resnet_cnsn.py:112-122, models/cifar/wideresnet_cnsn.py:86-96).
// Same thread mapping and streaming discipline as cnsn_stream_kernels.h; what changes is
//   ADD_PRE : the op's input is formed in registers as x + addend in every pass (never materialised;
//             rounded to the tensor's type like the reference's in-place `out += identity`),
//   ADD_POST: addend joins on the way out of the forward apply,
//   relu    : max(.,0) on the way out; the backward re-evaluates the forward affine with the very
//             coefficients the forward used (kept in `saved`) to recover the mask instead of reading y.
// HBM passes: forward 2+3 (PRE) or 1+3 (POST) or 1+2 (ReLU only) tensor passes against 3+3+2 for the
// unfused add -> CNSN -> ReLU; backward 3+4 against 3+5.
#pragma once
#include "cnsn_layout.h"
#include "cnsn_stream_kernels.h"

namespace cnsn {

enum AddMode { ADD_NONE = 0, ADD_PRE = 1, ADD_POST = 2 };

// stream three planes in lock step: consume(vecA, vecB, vecC, vec_index)
template <typename T, int VEC, int LPP, typename Consume>
__device__ __forceinline__ void stream3(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                        int nvec, int lane, Consume&& consume) {
    constexpr int U = 2;
    int i = lane;
    for (; i + (U - 1) * LPP < nvec; i += U * LPP) {
        Vec<T, VEC> va[U], vb[U], vc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = (size_t)(i + u * LPP) * VEC;
            va[u] = load_vec_nt<T, VEC>(a + o);
            vb[u] = load_vec_nt<T, VEC>(b + o);
            vc[u] = load_vec_nt<T, VEC>(c + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) consume(va[u], vb[u], vc[u], i + u * LPP);
    }
    for (; i < nvec; i += LPP) {
        const size_t o = (size_t)i * VEC;
        const Vec<T, VEC> va = load_vec_nt<T, VEC>(a + o), vb = load_vec_nt<T, VEC>(b + o), vc = load_vec_nt<T, VEC>(c + o);
        consume(va, vb, vc, i);
    }
}

// the forward's apply coefficients of one plane, as the forward kept them
struct FwdAffine {
    float a_in, xr, b_in, a_out, b_out;
    __device__ __forceinline__ FwdAffine(const double* __restrict__ saved, SvRec p) {
        a_in = (float)saved[sv_at(p, SV_FC0 + FC_A_IN)];
        xr = (float)saved[sv_at(p, SV_FC0 + FC_XR)];
        b_in = (float)saved[sv_at(p, SV_FC0 + FC_B_IN)];
        a_out = (float)saved[sv_at(p, SV_FC0 + FC_A_OUT)];
        b_out = (float)saved[sv_at(p, SV_FC0 + FC_B_OUT)];
    }
    __device__ __forceinline__ float in(float x) const { return fmaf(a_in, x - xr, b_in); }
    __device__ __forceinline__ float out(float x) const { return fmaf(a_out, x, b_out); }
};

// x + addend as the reference's `out += identity` leaves it: a value of the tensor's own type
template <typename T>
__device__ __forceinline__ float sum_t(float a, float b) {
    if constexpr (sizeof(T) == 4)
        return a + b;
    else
        return to_float(from_float<T>(a + b));
}

// is the stored output element positive?  (what nn.ReLU's backward tests; for 16-bit tensors the value is
// rounded the way the forward stored it first)
template <typename T>
__device__ __forceinline__ bool relu_open(float t) {
    if constexpr (sizeof(T) == 4)
        return t > 0.f;
    else
        return to_float(from_float<T>(t)) > 0.f;
}

// ------------------------------------------------------------------------------------------------
// pass A with the sum formed on the fly (ADD_PRE): moments of x + addend, layout as plane_stats_kernel
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED>
__global__ __launch_bounds__(kBlock) void fused_stats_kernel(const T* __restrict__ x, const T* __restrict__ addend,
                                                             Geom g, double* __restrict__ mom) {
    constexpr int NACC = BOXED ? 6 : 2;
    __shared__ float lds[4 * NACC];
    const PlaneId<LPP> id(g.P);
    const size_t off = (size_t)id.p * g.M;
    float K = 0.f;
    {
        const Vec<T, VEC> fa = load_vec<T, VEC>(x + off), fb = load_vec<T, VEC>(addend + off);
#pragma unroll
        for (int j = 0; j < VEC; ++j) K += sum_t<T>(to_float(fa.v[j]), to_float(fb.v[j]));
        K *= (1.0f / VEC);
    }
    float part[NACC][VEC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) part[k][j] = 0.f;
    stream2<T, VEC, LPP, true>(x + off, addend + off, g.nvec, id.lane,
                               [&](const Vec<T, VEC>& va, const Vec<T, VEC>& vb, int i) {
                                   const int e = i * VEC;
                                   const int r = BOXED ? e / g.Wd : 0, c = BOXED ? e - r * g.Wd : 0;
#pragma unroll
                                   for (int j = 0; j < VEC; ++j) {
                                       const float d = sum_t<T>(to_float(va.v[j]), to_float(vb.v[j])) - K;
                                       if constexpr (!BOXED) {
                                           part[0][j] += d;
                                           part[1][j] = fmaf(d, d, part[1][j]);
                                       } else {
                                           const float d2 = d * d;
                                           const bool ic = g.cb.has(r, c + j), is = g.sb.has(r, c + j);
                                           part[0][j] += ic ? d : 0.f;
                                           part[1][j] += ic ? d2 : 0.f;
                                           part[2][j] += ic ? 0.f : d;
                                           part[3][j] += ic ? 0.f : d2;
                                           part[4][j] += is ? d : 0.f;
                                           part[5][j] += is ? d2 : 0.f;
                                       }
                                   }
                               });
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

// ------------------------------------------------------------------------------------------------
// pass B with the epilogue: y = act(affine(x [+ addend]) [+ addend])
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED, int ADD>
__global__ __launch_bounds__(kBlock) void fused_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ addend,
                                                                 T* __restrict__ y, Geom g, ApplyCoef cf, int relu) {
    const PlaneId<LPP> id(g.P);
    if (!id.valid) return;
    const size_t off = (size_t)id.p * g.M;
    const float a_in = cf.a_in[id.p], xr = cf.xr[id.p], b_in = cf.b_in[id.p];
    const float a_out = cf.a_out[id.p], b_out = cf.b_out[id.p];
    T* yb = y + off;
    auto emit = [&](const Vec<T, VEC>& vx, const Vec<T, VEC>& vb, int i) {
        Vec<T, VEC> o;
        const int e = i * VEC;
        const int r = BOXED ? e / g.Wd : 0, c = BOXED ? e - r * g.Wd : 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float f = to_float(vx.v[j]);
            if constexpr (ADD == ADD_PRE) f = sum_t<T>(f, to_float(vb.v[j]));
            float t = (!BOXED || g.cb.has(r, c + j)) ? fmaf(a_in, f - xr, b_in) : fmaf(a_out, f, b_out);
            if constexpr (ADD == ADD_POST) t += to_float(vb.v[j]);
            o.v[j] = from_float<T>(relu ? fmaxf(t, 0.f) : t);
        }
        store_vec_nt<T, VEC>(yb + (size_t)i * VEC, o);
    };
    if constexpr (ADD == ADD_NONE)
        stream1<T, VEC, LPP, true>(x + off, g.nvec, id.lane, [&](const Vec<T, VEC>& v, int i) { emit(v, v, i); });
    else
        stream2<T, VEC, LPP, true>(x + off, addend + off, g.nvec, id.lane, emit);
}

// the (masked) upstream gradient and the op's input of one element
template <typename T, int ADD, bool BOXED>
__device__ __forceinline__ void masked_pair(float Gin, float xin, float bin, bool ic, const FwdAffine& fa, int relu,
                                            float& G, float& X) {
    X = ADD == ADD_PRE ? sum_t<T>(xin, bin) : xin;
    G = Gin;
    if (relu) {
        float t = (!BOXED || ic) ? fa.in(X) : fa.out(X);
        if (ADD == ADD_POST) t += bin;
        G = relu_open<T>(t) ? Gin : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// pass A' with the epilogue: sums of the masked gradient against x [+ addend]; rows as bwd_reduce_kernel
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED, int ADD>
__global__ __launch_bounds__(kBlock) void fused_bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                  const T* __restrict__ addend, Geom g,
                                                                  const double* __restrict__ saved, int relu,
                                                                  float* __restrict__ out) {
    constexpr int NACC = BOXED ? 4 : 2;
    __shared__ float lds[4 * NACC];
    const PlaneId<LPP> id(g.P);
    const size_t off = (size_t)id.p * g.M;
    const SvRec ps = sv_rec_of_plane((size_t)id.p, g.N, g.C);
    const float si = (float)saved[sv_at(ps, SV_MU_C)];
    const float so = BOXED ? (float)saved[sv_at(ps, SV_MU_O)] : 0.f;
    const FwdAffine fa(saved, ps);  // rows hold garbage without ReLU; never used then
    float part[NACC][VEC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) part[k][j] = 0.f;
    auto eat = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, const Vec<T, VEC>& vb, int i) {
        const int e = i * VEC;
        const int r = BOXED ? e / g.Wd : 0, c = BOXED ? e - r * g.Wd : 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool ic = !BOXED || g.cb.has(r, c + j);
            float G, X;
            masked_pair<T, ADD, BOXED>(to_float(vg.v[j]), to_float(vx.v[j]), to_float(vb.v[j]), ic, fa, relu, G, X);
            if constexpr (!BOXED) {
                part[0][j] += G;
                part[1][j] = fmaf(G, X - si, part[1][j]);
            } else {
                part[0][j] += ic ? G : 0.f;
                part[1][j] += ic ? G * (X - si) : 0.f;
                part[2][j] += ic ? 0.f : G;
                part[3][j] += ic ? 0.f : G * (X - so);
            }
        }
    };
    if constexpr (ADD == ADD_NONE)
        stream2<T, VEC, LPP, true>(gy + off, x + off, g.nvec, id.lane,
                                   [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, int i) { eat(vg, vx, vx, i); });
    else
        stream3<T, VEC, LPP>(gy + off, x + off, addend + off, g.nvec, id.lane, eat);
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
    if (id.lane == 0 && id.valid) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) out[(size_t)k * g.P + id.p] = acc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// pass B' with the epilogue: dx from the masked gradient (coefficients as apply_bwd_kernel);
// ADD_POST + ReLU also writes the masked gradient itself (= gradient of addend)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPP, bool BOXED, int ADD>
__global__ __launch_bounds__(kBlock) void fused_apply_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                                 const T* __restrict__ addend, T* __restrict__ dx,
                                                                 T* __restrict__ d_addend, Geom g,
                                                                 const float* __restrict__ coef,
                                                                 const double* __restrict__ saved, int relu) {
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
    const FwdAffine fa(saved, sv_rec_of_plane((size_t)id.p, g.N, g.C));
    T* db = dx + off;
    auto emit = [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, const Vec<T, VEC>& vb, int i) {
        Vec<T, VEC> o, om;
        const int e = i * VEC;
        const int r = BOXED ? e / g.Wd : 0, c = BOXED ? e - r * g.Wd : 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool ic = !BOXED || g.cb.has(r, c + j);
            float G, X;
            masked_pair<T, ADD, BOXED>(to_float(vg.v[j]), to_float(vx.v[j]), to_float(vb.v[j]), ic, fa, relu, G, X);
            float d;
            if constexpr (!BOXED) {
                d = fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i));
            } else {
                d = ic ? fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i)) : fmaf(cG_o, G, fmaf(cX_o, X - xr_o, c0_o));
                d += g.sb.has(r, c + j) ? fmaf(eS, X - xs, e0) : 0.f;
            }
            o.v[j] = from_float<T>(d);
            om.v[j] = from_float<T>(G);
        }
        store_vec_nt<T, VEC>(db + (size_t)i * VEC, o);
        if constexpr (ADD == ADD_POST) {
            if (d_addend) store_vec_nt<T, VEC>(d_addend + off + (size_t)i * VEC, om);
        }
    };
    if constexpr (ADD == ADD_NONE)
        stream2<T, VEC, LPP, true>(gy + off, x + off, g.nvec, id.lane,
                                   [&](const Vec<T, VEC>& vg, const Vec<T, VEC>& vx, int i) { emit(vg, vx, vx, i); });
    else
        stream3<T, VEC, LPP>(gy + off, x + off, addend + off, g.nvec, id.lane, emit);
}

}  // namespace cnsn
