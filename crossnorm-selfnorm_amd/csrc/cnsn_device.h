// Device-side building blocks shared by every kernel of libcnsn_hip.so (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnsn {

constexpr int kBlock = 256;  // 4 waves of 64 lanes

// ------------------------------------------------------------------------------------------------
// element types: float, bf16 (own bit-level type), _Float16
// ------------------------------------------------------------------------------------------------
struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(bf16_t v) { return __uint_as_float(uint32_t(v.bits) << 16); }
__device__ __forceinline__ float to_float(_Float16 v) { return float(v); }

template <typename T>
__device__ __forceinline__ T from_float(float f);
template <>
__device__ __forceinline__ float from_float<float>(float f) {
    return f;
}
template <>
__device__ __forceinline__ bf16_t from_float<bf16_t>(float f) {
    // gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN)
    bf16_t r;
    r.bits = __builtin_bit_cast(uint16_t, (__bf16)f);
    return r;
}
template <>
__device__ __forceinline__ _Float16 from_float<_Float16>(float f) {
    return _Float16(f);
}

// VEC consecutive elements moved by ONE global_load / global_store of VEC*sizeof(T) bytes
// (16 B per lane whenever the plane size allows it: 1 KiB per wave-instruction).
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Vec {
    T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ Vec<T, VEC> load_vec(const T* p) {
    return *reinterpret_cast<const Vec<T, VEC>*>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const Vec<T, VEC>& v) {
    *reinterpret_cast<Vec<T, VEC>*>(p) = v;
}

// streaming variants (nt = non-temporal hint): every tensor here is far larger than the caches and each
// line is used once per kernel, so keeping it resident only evicts something useful.  Measured on the
// two-pass kernels at (256,256,56,56) fp32: apply_fwd 337 -> 262 us, apply_bwd 498 -> 418 us.
#ifndef CNSN_NO_NT
#define CNSN_NT_LOAD 1
#define CNSN_NT_STORE 1
#endif
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
template <typename T, int VEC>
__device__ __forceinline__ Vec<T, VEC> load_vec_nt(const T* p) {
#ifdef CNSN_NT_LOAD
    if constexpr (sizeof(T) * VEC == 16)
        return __builtin_bit_cast(Vec<T, VEC>, __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p)));
    else if constexpr (sizeof(T) * VEC == 8)
        return __builtin_bit_cast(Vec<T, VEC>, __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p)));
    else
#endif
        return load_vec<T, VEC>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec_nt(T* p, const Vec<T, VEC>& v) {
#ifdef CNSN_NT_STORE
    if constexpr (sizeof(T) * VEC == 16)
        __builtin_nontemporal_store(__builtin_bit_cast(nt_u4, v), reinterpret_cast<nt_u4*>(p));
    else if constexpr (sizeof(T) * VEC == 8)
        __builtin_nontemporal_store(__builtin_bit_cast(nt_u2, v), reinterpret_cast<nt_u2*>(p));
    else
#endif
        store_vec<T, VEC>(p, v);
}

// ------------------------------------------------------------------------------------------------
// cross-lane sums.  DPP inside a 16-lane row (no LDS traffic), v_readlane across the four rows.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// every lane of each 16-lane row ends with that row's sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}

// the wave's sum as the raw bits of lane 63 (an SGPR): the four row sums are folded with two row-broadcast DPP adds —
// every row takes the last lane of the row before it, then rows 2 and 3 take lane 31: lane 63 = (R2 + R3) + (R0 + R1), the
// same two-level order as reading the four row sums and adding them pairwise (bit-identical), in 3 instructions instead
// of 7.  (All rows enabled and bound_ctrl on, so that the DPP move folds into the add; the other rows end with sums nobody
// reads.)
__device__ __forceinline__ int wave_sum_bits(float v) {
    v = row16_sum(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xF, 0xF, true));  // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xF, 0xF, true));  // row_bcast:31
    return __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63);
}
// every lane ends with the wave's sum (wave-uniform)
__device__ __forceinline__ float wave_sum(float v) { return __builtin_bit_cast(float, wave_sum_bits(v)); }
// lane LANE of `old` replaced by the wave-uniform `sval` (no builtin in this hipcc).  An SGPR source operand of
// v_writelane has no wait-state requirement after the v_readlane that produced it (only a lane SELECT would)
template <int LANE>
__device__ __forceinline__ int put_lane(int sval, int old) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(LANE));
    return old;
}

// Sum NACC accumulators over the LPP lanes that share a plane (LPP = 16, 64 or 256 = whole block).
// `lds` needs 4*NACC floats when LPP == 256.  Every participating lane gets the result.
template <int LPP, int NACC>
__device__ __forceinline__ void group_sum(float (&acc)[NACC], float* lds) {
    if constexpr (LPP == 16) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = row16_sum(acc[k]);
    } else if constexpr (LPP == 64) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = wave_sum(acc[k]);
    } else {
        static_assert(LPP == 256, "lanes per plane must be 16, 64 or 256");
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = wave_sum(acc[k]);
        __syncthreads();  // lds may still be read from a previous use
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) lds[wave * NACC + k] = acc[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NACC; ++k)
            acc[k] = (lds[k] + lds[NACC + k]) + (lds[2 * NACC + k] + lds[3 * NACC + k]);
    }
}

// block-wide sum of NACC doubles through LDS (mid kernels; one block per channel)
template <int NACC>
__device__ __forceinline__ void block_sum_d(double (&acc)[NACC], double* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        for (int k = 0; k < NACC; ++k) lds[wave * NACC + k] = acc[k];
    }
    __syncthreads();
#ifdef CNSN_BLOCKSUM_WIDE
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double s = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) s += lds[w * NACC + k];
        acc[k] = s;
    }
#else
    // lane l reads the partial of wave l % 4 and every group of four lanes adds them up: NACC doubles live per lane
    // instead of 4 * NACC (these sums run while whole planes sit in the caller's registers)
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double v = lds[(lane & 3) * NACC + k];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        acc[k] = v;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// geometry of one launch over planes
// ------------------------------------------------------------------------------------------------
struct Box {
    int r0, c0, r1, c1;  // rows [r0,r1) of dim 2, columns [c0,c1) of dim 3
    __host__ __device__ int area() const { return (r1 - r0) * (c1 - c0); }
    __device__ __forceinline__ bool has(int r, int c) const {
        return (unsigned)(r - r0) < (unsigned)(r1 - r0) && (unsigned)(c - c0) < (unsigned)(c1 - c0);
    }
};

struct Geom {
    int P;     // planes = N*C
    int N, C;  // (plane p = n*C + c; the `saved` records are channel-major: cnsn_layout.h)
    int M;     // elements per plane = H*W
    int Wd;    // width (dim 3)
    int nvec;  // M / VEC
    Box cb;    // content box (whole plane when the call has none)
    Box sb;    // style box   (whole plane when the call has none)
    int keep;  // first pass of a two-pass direction with the default cache policy (small tensors: the second pass
               // finds them in L2 / the Infinity Cache); 0 = non-temporal like every other plane access
};

}  // namespace cnsn
