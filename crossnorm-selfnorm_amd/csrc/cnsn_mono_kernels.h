// Channel-in-registers strategy ("mono") for SMALL planes without CrossNorm: ONE 1024-thread workgroup per channel,
// every plane of the channel held in the workgroup's VGPRs, one launch per direction, single touch, and no
// inter-workgroup exchange of any kind.
//
// Where it sits between the other two single-touch strategies:
//   cluster-resident (cnsn_resident_kernels.h): a channel spread over K co-resident workgroups + an exchange through
//       memory.  Right for planes of several KiB; for a 392-byte plane (14x14 bf16) the exchange costs more than moving
//       the planes — (256,1024,14,14) bf16 ran at 37 % of its HBM bound in round 1.
//   channel-local (cnsn_local_kernels.h): a channel (group) staged in ONE workgroup's LDS.  Right for 7x7 / 8x8 (planes
//       that are not a whole number of 8-byte vectors); a 100 KiB image leaves one workgroup per CU and every phase
//       goes through LDS twice.
//   mono (this file): a plane of at most 64 vectors (8 or 16 bytes each) is ONE register slot of LPP = 16 or 64 lanes;
//       a wave holds R slot rows, the 16 waves of the workgroup hold all N planes of the channel:
//       N * M * b <= 16 waves * 64 lanes * R * vector bytes  — (256,.,14,14): R = 16 slots of 8 bytes (bf16) or 16 bytes
//       (fp32) per lane.  Plane statistics are DPP sums inside the slot's lane group (exact two-pass from registers),
//       SelfNorm's BatchNorm1d over N is one thread per plane + one workgroup reduction, the apply runs from registers.
// Algebra and `saved` contract are those of the channel-local kernels (SelfNorm alone, optional residual-block
// epilogue: PRE add and ReLU), so a forward of this strategy can be followed by a backward of any other.
#pragma once
#include "../../include/cnsn_hip.h"
#include "cnsn_algebra.h"
#include "cnsn_device.h"
#include "cnsn_fused_stream_kernels.h"
#include "cnsn_layout.h"
#include "cnsn_resident_kernels.h"  // Raw / elem / pack / buf_load / buf_store / relu_open_r / add_raw

namespace cnsn {

// A slot of the mono kernels is VEC elements = 16, 8, 4 or 2 bytes per lane.  The two narrow forms serve planes that are
// not a whole number of 8-byte vectors — 7x7: one element per lane, 49 of 64 lanes, a wave-wide access is the plane's 98
// (bf16) or 196 (fp32) contiguous bytes; neighbouring channels, contiguous in memory, are neighbouring workgroups on one XCD.
template <int BYTES>
struct MRawOf;
template <>
struct MRawOf<16> {
    using type = v4i_t;
};
template <>
struct MRawOf<8> {
    using type = v2i_t;
};
template <>
struct MRawOf<4> {
    using type = int;
};
template <>
struct MRawOf<2> {
    using type = unsigned short;
};
template <typename T, int VEC>
using MRaw = typename MRawOf<(int)sizeof(T) * VEC>::type;

template <typename T, int VEC>
__device__ __forceinline__ float melem(const MRaw<T, VEC>& r, int q) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B >= 8) {
        return elem<T, VEC>(r, q);
    } else if constexpr (sizeof(T) == 4) {  // one float
        return __int_as_float(r);
    } else {
        unsigned short h;
        if constexpr (B == 4)
            h = (q & 1) ? (unsigned short)((unsigned)r >> 16) : (unsigned short)((unsigned)r & 0xffffu);
        else
            h = r;
        if constexpr (__is_same(T, bf16_t))
            return __uint_as_float((unsigned)h << 16);
        else
            return (float)__builtin_bit_cast(_Float16, h);
    }
}
template <typename T>
__device__ __forceinline__ unsigned short mhalf_bits(float f) {
    if constexpr (__is_same(T, bf16_t))
        return from_float<bf16_t>(f).bits;
    else
        return __builtin_bit_cast(unsigned short, from_float<_Float16>(f));
}
template <typename T, int VEC>
__device__ __forceinline__ MRaw<T, VEC> mpack(const float (&f)[VEC]) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B >= 8) {
        return pack<T, VEC>(f);
    } else if constexpr (sizeof(T) == 4) {
        return __float_as_int(f[0]);
    } else if constexpr (B == 4) {
        return (int)(((unsigned)mhalf_bits<T>(f[1]) << 16) | (unsigned)mhalf_bits<T>(f[0]));
    } else {
        return mhalf_bits<T>(f[0]);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ MRaw<T, VEC> mload(__amdgpu_buffer_rsrc_t r, int voff) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B >= 8)
        return buf_load<T, VEC>(r, voff);
    else if constexpr (B == 4)
        return __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, CNSN_RES_LOAD_AUX);
    else
        return __builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, CNSN_RES_LOAD_AUX);
}
// the same load with the default cache policy (no non-temporal hint): what a kernel reads TWICE (mono_bwd_kernel, RELOAD)
template <typename T, int VEC>
__device__ __forceinline__ MRaw<T, VEC> mload_cached(__amdgpu_buffer_rsrc_t r, int voff) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B == 16)
        return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    else if constexpr (B == 8)
        return __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    else if constexpr (B == 4)
        return __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
    else
        return __builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0);
}
template <typename T, int VEC>
__device__ __forceinline__ void mstore(__amdgpu_buffer_rsrc_t r, int voff, const MRaw<T, VEC>& v) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B >= 8)
        buf_store<T, VEC>(r, voff, v);
    else if constexpr (B == 4)
        __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, 0, CNSN_RES_STORE_AUX);
    else
        __builtin_amdgcn_raw_buffer_store_b16(v, r, voff, 0, CNSN_RES_STORE_AUX);
}
template <typename T, int VEC>
__device__ __forceinline__ MRaw<T, VEC> madd(const MRaw<T, VEC>& a, const MRaw<T, VEC>& b) {
    float f[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) f[q] = melem<T, VEC>(a, q) + melem<T, VEC>(b, q);
    return mpack<T, VEC>(f);
}

#ifndef MONO_ILP
#define MONO_ILP 2  // slot rows the scheduler may interleave in the backward's slot loops (power of two)
#endif

// waves per SIMD asked of the compiler: a 1024-thread workgroup is 4 per SIMD; 8 = TWO workgroups per CU (64 VGPRs), so
// that one workgroup's loads and stores overlap the other's statistics / algebra — possible when a tensor's slot rows
// take at most 32 registers (the forward of 16-bit (256,.,14,14), of fp32 up to 128 planes per channel)
#ifndef MONO_FWD_WAVES_SMALL
#define MONO_FWD_WAVES_SMALL 8   // measured: block forward (256,1024,14,14) bf16 0.096 -> 0.078 ms, plain forward unchanged
#endif
constexpr int mono_fwd_waves(int data_regs) { return data_regs <= 32 ? MONO_FWD_WAVES_SMALL : 4; }

constexpr int kMonoBlock = 1024;
constexpr int kMonoWaves = kMonoBlock / 64;

struct MonoArgs {
    MidArgs mid;
    int nvec;  // vectors per plane (<= LPP)
    int R;     // slot rows in use per wave (<= RMAX): N <= 16 * R * (64 / LPP)
};

// cap = plane slots of the workgroup = 16 waves * R rows * (64 / LPP) planes per row (>= N): the per-plane LDS arrays are
// that long, so a slot row past the batch end indexes real (unused) LDS and needs no clamp
// (tail: the BatchNorm2d + ReLU variant keeps one / two more floats per plane)
__host__ __device__ inline size_t mono_lds_bytes(int cap, bool backward, bool tail = false) {
    const size_t n = (size_t)cap;
    return (backward ? (tail ? 9 : 7) * n * 4 + 5 * n * 8 : (tail ? 3 : 2) * n * 4) + (size_t)kMonoWaves * 4 * 8 + 16 * 8;
}

// TAIL variant of the two kernels (second half): the NEXT block's `relu(bn1(.))` evaluated in the same launch
// — models/cifar/wideresnet_cnsn.py:93-96 (`out = torch.add(x, out); return self.cnsn(out)`) followed by :69-70 / :76-77
// (`self.relu1(self.bn1(x))`) and, after the last block, :222.  The whole channel sits in this workgroup, and SelfNorm's
// output is y = g[n] * X per plane, so BatchNorm2d's batch statistics over (N, H, W) follow from the plane moments already
// on chip: mean = sum_n g mu / N, E[y^2] = sum_n g^2 (M2/M + mu^2) / N; z = relu(A[n] * X + B) with A = gamma2*rstd2*g,
// B = beta2 - gamma2*rstd2*mean2 is a second per-plane affine of the registers — one more store stream, no extra read.
// Backward: both gradients meet in H = Gy + gamma2*rstd2 * (Gz where z > 0), formed on the way in (gamma2, rstd2 and the
// mask's coefficients are known from the forward), so the kernel still holds two tensors; the BatchNorm2d backward's two
// channel sums come from two more per-plane sums taken in the same pass, and its x-dependent part folds into the slope
// of dx (csrc/cnsn_mono_kernels.h, "tail" blocks; algebra in DESIGN.md §4.7).
struct TailDev {
    const float* weight;   // (C) BatchNorm2d weight, bias
    const float* bias;
    float* run_mean;       // (C) updated in place when training
    float* run_var;
    float* stats;          // (2, C) batch mean and rstd the forward used: written forward, read backward
    float* d_weight;       // (C) backward
    float* d_bias;
    void* z;               // forward: second output, same shape as y
    const void* gz;        // backward: gradient of z
    float eps, momentum;
    int training;
    long long* nbt;        // BatchNorm2d.num_batches_tracked or null: += 1 by the forward when training
};
// z-coefficients of one plane — ONE definition for the forward and the backward's mask
__device__ __forceinline__ void tail_coefs(float g, float gamma2, float beta2, double m2, double r2, float& A, float& B) {
    const double k = (double)gamma2 * r2;
    A = (float)(k * (double)g);
    B = (float)((double)beta2 - k * m2);
}

// sum of NACC doubles over the whole workgroup (every thread calls it); red: [kMonoWaves][NACC]
template <int NACC>
__device__ __forceinline__ void mono_block_sum(double (&acc)[NACC], double* red) {
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
        for (int k = 0; k < NACC; ++k) red[wave * NACC + k] = acc[k];
    }
    __syncthreads();
    // the 16 partials: lane l reads the one of wave l % 16 and the four 16-lane groups of a wave each reduce to the
    // total (4 values per lane; reading all 64 partials into every lane costs 128 VGPRs while planes sit in registers)
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double v = red[(lane & 15) * NACC + k];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[k] = v;
    }
}

// Tell the compiler that the packed registers of a slot "changed" (they did not): values unpacked from them before this
// point cannot be kept alive across it, so the phases of a kernel re-unpack the 16-bit / re-read the 32-bit lanes of
// the slot instead of holding RMAX * VEC floats per tensor next to the packed planes (which spilled to scratch).
template <typename V>
__device__ __forceinline__ void mono_forget(V& v) {
    if constexpr (sizeof(V) >= 8) {
        constexpr int W = (int)(sizeof(V) / 4);
#pragma unroll
        for (int i = 0; i < W; ++i) {
            int t = v[i];
            asm volatile("" : "+v"(t));
            v[i] = t;
        }
    } else {
        int t = (int)v;
        asm volatile("" : "+v"(t));
        v = (V)t;
    }
}

// Channel order.  Workgroups are dealt to the 8 XCDs round-robin (b % 8) and every XCD has its own L2: giving each XCD
// a CONTIGUOUS range of channels makes the workgroups that run side by side on one XCD (b, b + 8, ...) work on ADJACENT
// channels.  Their planes share 128-byte lines at both ends (a 392-byte plane is 3.06 lines), which then meet in one
// L2 — read once, written as whole lines — instead of travelling to two XCDs: measured +8 % at (256,1024,14,14) bf16.
// The grid is C rounded up to a multiple of 8 workgroups; workgroup b takes channel start_x + b / 8 of its XCD's range
// (x = b % 8), the few workgroups past the end of a range leave at once.  (A persistent walk — G8 workgroups per XCD
// stepping through the range, LDS state in two alternating copies — was measured and dropped: the loop cost 20-70
// spilled VGPRs and the forward got slower, 0.082 -> 0.088 ms at (256,1024,14,14) bf16, 0.095 -> 0.120 ms fp32.)
// Tuning builds (-DMONO_SKEW=steps of s_sleep(127), -DMONO_SKEW_MODE): every workgroup of a launch starts its load phase at
// the same moment, computes while the memory system idles and stores together again; delaying every other workgroup of a CU
// was tried as a way to interleave the phases of the two that share it.  MEASURED, NOT KEPT (profiles/r03_sn_cluster.md §8):
// (256,1024,14,14) bf16 forward / backward 0.063 / 0.097 ms -> 0.066-0.079 / 0.104-0.115 with 2-4 steps on any of the three
// partitions, 7x7 unchanged or slower: the delay only lengthens the tail of a launch that has two rounds of workgroups.
#ifndef MONO_SKEW
#define MONO_SKEW 0
#endif
#ifndef MONO_SKEW_MODE
#define MONO_SKEW_MODE 0
#endif
__device__ __forceinline__ void mono_startup_skew() {
#if MONO_SKEW > 0
    const bool late = MONO_SKEW_MODE == 0 ? ((blockIdx.x >> 8) & 1) : (MONO_SKEW_MODE == 1 ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1));
    if (late)
        for (int i = 0; i < MONO_SKEW; ++i) __builtin_amdgcn_s_sleep(127);
#endif
}

struct MonoWalk {
    int start, count, j;
    __device__ __forceinline__ MonoWalk(int C) {
        const int b = blockIdx.x, x = b & 7, per = C >> 3, rem = C & 7;
        start = x * per + (x < rem ? x : rem);
        count = per + (x < rem ? 1 : 0);
        j = b >> 3;
    }
};

template <int LPP>
__device__ __forceinline__ float mono_group_sum(float v) {
    if constexpr (LPP == 16)
        return row16_sum(v);
    else
        return wave_sum(v);
}

// slot geometry of one lane: plane row `wave * R + r` holds planes (wave * R + r) * PPR + sub, vector vl of each
template <typename T, int VEC, int LPP>
struct MonoGeom {
    static constexpr int PPR = 64 / LPP;
    static constexpr int VB = VEC * (int)sizeof(T);
    int sub, vl, wave, C, M, R;
    long long e0;  // element offset of slot row 0 of this wave in channel c (wave-uniform, kept in SGPRs)
    int stride;    // elements between consecutive slot rows = PPR * C * M
    int rlim;  // this lane holds real data in slot rows r < rlim (0 for a lane past the end of the plane): ONE register
               // decides validity of every row — r < rlim is a compare against a constant after unrolling
    int voff;  // byte offset of this lane's vector inside its slot row
    __device__ __forceinline__ MonoGeom(const MonoArgs& ma) {
        const int lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        sub = lane / LPP;
        vl = lane - sub * LPP;
        C = ma.mid.C;
        M = ma.mid.M;
        R = ma.R;
        // plane(r) = (wave * R + r) * PPR + sub < N  <=>  r < ceil((N - sub) / PPR) - wave * R
        int lim = (ma.mid.N - sub + PPR - 1) / PPR - wave * R;
        lim = lim < 0 ? 0 : (lim > R ? R : lim);
        rlim = vl < ma.nvec ? lim : 0;
        voff = (sub * C * M + vl * VEC) * (int)sizeof(T);
        stride = PPR * C * M;
        e0 = 0;
    }
    // channel of this workgroup: fixes the (scalar) element offset of the wave's first slot row
    __device__ __forceinline__ void set_channel(int c) {
        const long long v = ((long long)(wave * R * PPR) * C + c) * M;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
        e0 = (long long)(((unsigned long long)hi << 32) | lo);
    }
    // Between the phases of a kernel: make the compiler forget what it derived from the lane geometry, so that the
    // per-row offsets are recomputed (one compare + select each) instead of being kept — or spilled — across phases.
    __device__ __forceinline__ void refresh() {
        asm volatile("" : "+v"(rlim), "+v"(voff));
    }
    // (rows r >= R are never ok(): their loads return zeros without touching memory and their stores are dropped, so the
    //  slot loops below run over all RMAX rows without control flow — the register arrays stay in registers)
    __device__ __forceinline__ int plane(int r) const { return (wave * R + r) * PPR + sub; }
    __device__ __forceinline__ bool ok(int r) const { return r < rlim; }
    // descriptor of slot row r of channel c of tensor `t`: lanes that are not ok() get an offset past its end
    // (the base must be wave-uniform for the compiler too — e0 went through readfirstlane —: a descriptor built in
    //  VGPRs wraps every access in a waterfall loop)
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const T* t, int /*c*/, int r) const {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(t + e0 + (long long)r * stride), 0, 0x7ffffff0, 0x00020000);
    }
    __device__ __forceinline__ int off(int r) const { return ok(r) ? voff : 0x7ffffff8; }
};

// ================================================================================================
// forward
// ================================================================================================
template <typename T, int VEC, int LPP, int RMAX, bool EPI, bool TAIL = false>
__global__ __launch_bounds__(kMonoBlock, TAIL ? 4 : mono_fwd_waves(RMAX * VEC * (int)sizeof(T) / 4)) void mono_fwd_kernel(MonoArgs ma, const T* __restrict__ x,
                                                                  const T* __restrict__ addend, T* __restrict__ y, GateDev gg,
                                                                  GateDev gf, double* __restrict__ saved, int add, int relu, TailDev tl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MidArgs a = ma.mid;
    a.sn_two = 0;  // (the host does not send the two-gate form here)
    const int N = a.N, C = a.C;
    const int npad = kMonoWaves * ma.R * (64 / LPP);
    const MonoWalk wk(C);
    mono_startup_skew();
    if (wk.j < wk.count) {
    const int c = wk.start + wk.j;
    char* lds = smem;
    float* pmu = (float*)lds;  // [N] mean, later slope
    float* pm2 = pmu + npad;    // [N] M2, later offset
    float* pza = pm2 + (TAIL ? npad : 0);  // [N] TAIL: slope of z
    double* red = (double*)(pza + npad);
    double* par = red + kMonoWaves * 4;  // [16] per-channel parameters
    const size_t P = (size_t)N * C;
    MonoGeom<T, VEC, LPP> g(ma);
    g.set_channel(c);

    if (threadIdx.x == 0) {  // per-channel parameters, fetched ahead of the bulk loads
        par[0] = gg.w[2 * c];
        par[1] = gg.w[2 * c + 1];
        par[2] = gg.gamma[c];
        par[3] = gg.beta[c];
        par[4] = gg.run_mean[c];
        par[5] = gg.run_var[c];
        if (a.sn_two) {
            par[6] = gf.w[2 * c];
            par[7] = gf.w[2 * c + 1];
            par[8] = gf.gamma[c];
            par[9] = gf.beta[c];
            par[10] = gf.run_mean[c];
            par[11] = gf.run_var[c];
        }
        if constexpr (TAIL) {
            par[12] = tl.weight[c];
            par[13] = tl.bias[c];
            par[14] = tl.run_mean[c];
            par[15] = tl.run_var[c];
        }
    }

    // ---- the only read of x (+ addend): every plane of the channel into registers
    MRaw<T, VEC> d[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        d[r] = mload<T, VEC>(g.rsrc(x, c, r), g.off(r));
    if constexpr (EPI) {
        if (add == ADD_PRE) {  // the op's input is x + addend (rounded to T like `out += identity`), never written
            constexpr int CH = (RMAX * VEC * (int)sizeof(T) > 128) ? RMAX / 2 : RMAX;  // bound the transient registers
#pragma unroll
            for (int r0 = 0; r0 < RMAX; r0 += CH) {
                MRaw<T, VEC> q[CH];
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    q[r] = mload<T, VEC>(g.rsrc(addend, c, r0 + r), g.off(r0 + r));
#pragma unroll
                for (int r = 0; r < CH; ++r)
                    d[r0 + r] = madd<T, VEC>(d[r0 + r], q[r]);
            }
        }
    }

    // ---- exact two-pass statistics of every plane, from registers
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) s += melem<T, VEC>(d[r], q);  // lanes that are not ok() loaded zeros
            const float mean = mono_group_sum<LPP>(s) / (float)a.M;
            float m2 = 0.f;
            mono_forget(d[r]);
            if (g.ok(r)) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float t = melem<T, VEC>(d[r], q) - mean;
                    m2 = fmaf(t, t, m2);
                }
            }
            m2 = mono_group_sum<LPP>(m2);
            if (g.vl == 0 && g.ok(r)) {  // (rows r >= R alias the planes of other waves: never written)
                pmu[g.plane(r)] = mean;
                pm2[g.plane(r)] = m2;
            }
        }
    }
    __syncthreads();

    // ---- gates: BatchNorm1d over the N planes of the channel, a thread per plane (cnsn.py:137-141)
    using Rr = float;
    const int n = threadIdx.x;
    const bool act = n < N;
    const double wg0 = par[0], wg1 = par[1], wf0 = par[6], wf1 = par[7];
    FwdPlaneT<Rr> f{};
    double zg = 0.0, zf = 0.0;
    if (act) {
        MomentsT<Rr> o;
        o.mu_c = o.mu_s = pmu[n];
        o.M2c = o.M2s = pm2[n];
        o.mu_o = o.M2o = 0.f;
        f = fwd_plane<Rr>(a, o, 0.f, 0.f);
        zg = wg0 * (double)f.mu_p + wg1 * (double)f.sig_p;
        zf = a.sn_two ? wf0 * (double)f.mu_p + wf1 * (double)f.sig_p : 0.0;
    }
    double mg = par[4], mf = par[10], rg, rf;
    if (a.sn_training) {
        // batch sums in double about no shift: |z| = O(10), N <= 1024 -> the variance is good to ~1e-13 absolute,
        // eight orders below eps_bn
        double sz[4] = {act ? zg : 0.0, act ? zg * zg : 0.0, act ? zf : 0.0, act ? zf * zf : 0.0};
        mono_block_sum<4>(sz, red);
        mg = sz[0] * a.inv_n;
        mf = sz[2] * a.inv_n;
        double vg = sz[1] * a.inv_n - mg * mg, vf = sz[3] * a.inv_n - mf * mf;
        vg = vg > 0.0 ? vg : 0.0;
        vf = vf > 0.0 ? vf : 0.0;
        rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
        rf = (double)__builtin_amdgcn_rsqf((float)(vf + (double)a.eps_bn));
        if (threadIdx.x == 0) {
            const double mom_ = a.momentum, unb = a.unbias_n;
            gg.run_mean[c] = (float)((1.0 - mom_) * par[4] + mom_ * mg);
            gg.run_var[c] = (float)((1.0 - mom_) * par[5] + mom_ * vg * unb);
            if (a.sn_two) {
                gf.run_mean[c] = (float)((1.0 - mom_) * par[10] + mom_ * mf);
                gf.run_var[c] = (float)((1.0 - mom_) * par[11] + mom_ * vf * unb);
            }
            if (c == 0) {
                bump_batches_tracked(gg.nbt);
                if (a.sn_two) bump_batches_tracked(gf.nbt);
            }
        }
    } else {
        rg = (double)__builtin_amdgcn_rsqf((float)par[5] + a.eps_bn);
        rf = a.sn_two ? (double)__builtin_amdgcn_rsqf((float)par[11] + a.eps_bn) : 1.0;
    }
    if (saved && threadIdx.x == 0) {
        saved[SV_ROWS * P + c] = rg;
        saved[SV_ROWS * P + C + c] = rf;
    }
    if (act) {
        const double zhg = (zg - mg) * rg;
        const Rr gt = sigmoid_r<Rr>((Rr)(par[2] * zhg + par[3]));
        double zhf = 0.0;
        Rr ft = 1.f;
        if (a.sn_two) {
            zhf = (zf - mf) * rf;
            ft = sigmoid_r<Rr>((Rr)(par[8] * zhf + par[9]));
        }
        const FwdCoefs cf = fwd_coefs<Rr>(a, f, gt, ft);
        if (saved) {
            const SvRec p = sv_rec(n, c, N);
            store_fwd_plane<Rr>(saved, P, p, f, 0);
            saved[sv_at(p, SV_G)] = gt;
            saved[sv_at(p, SV_ZH_G)] = zhg;
            saved[sv_at(p, SV_F)] = ft;
            saved[sv_at(p, SV_ZH_F)] = zhf;
            if (a.save_coefs) store_fwd_coefs(saved, p, cf);
        }
        if constexpr (!TAIL) {
            pmu[n] = cf.a_in;  // SelfNorm alone: y = a_in * x + b_in  (xr = 0)
            pm2[n] = cf.b_in;
        }
    }
    float zb = 0.f;
    if constexpr (TAIL) {
        // ---- BatchNorm2d over (N, H, W) of y = g * X (wideresnet_cnsn.py:76-77): batch statistics from the plane moments
        float gt_ = 0.f;
        double ty[2] = {0.0, 0.0};
        if (act) {
            const double zhg = (zg - mg) * rg;  // (the same expressions as above: the gate of this thread's plane)
            gt_ = sigmoid_r<Rr>((Rr)(par[2] * zhg + par[3]));
            const double mu = (double)f.mu_p, gd = (double)gt_;
            ty[0] = gd * mu;
            ty[1] = gd * gd * ((double)pm2[n] / (double)a.M + mu * mu);
        }
        double m2, r2;
        if (tl.training) {
            mono_block_sum<2>(ty, red);
            m2 = ty[0] * a.inv_n;
            double v2 = ty[1] * a.inv_n - m2 * m2;
            v2 = v2 > 0.0 ? v2 : 0.0;
            r2 = (double)__builtin_amdgcn_rsqf((float)(v2 + (double)tl.eps));
            if (threadIdx.x == 0) {
                const double cnt = (double)N * (double)a.M, mom_ = tl.momentum;
                tl.run_mean[c] = (float)((1.0 - mom_) * par[14] + mom_ * m2);
                tl.run_var[c] = (float)((1.0 - mom_) * par[15] + mom_ * v2 * (cnt / (cnt - 1.0)));
                if (c == 0) bump_batches_tracked(tl.nbt);
            }
        } else {
            m2 = par[14];
            r2 = (double)__builtin_amdgcn_rsqf((float)par[15] + tl.eps);
            __syncthreads();  // (pm2 is read above and rewritten below, as in the training branch's reduction)
        }
        if (threadIdx.x == 0 && tl.stats) {
            tl.stats[c] = (float)m2;
            tl.stats[C + c] = (float)r2;
        }
        // (r2 travels as a float: the backward rebuilds the SAME coefficients from tl.stats)
        const double m2f = (double)(float)m2, r2f = (double)(float)r2;
        float za = 0.f;
        tail_coefs(gt_, (float)par[12], (float)par[13], m2f, r2f, za, zb);
        if (act) {
            pmu[n] = gt_;  // y = g * X
            pm2[n] = 0.f;
            pza[n] = za;
            if (saved) {   // the backward's mask re-evaluates z = A * X + B with exactly these numbers
                const SvRec p = sv_rec(n, c, N);
                saved[sv_at(p, SV_FC0 + FC_A_IN)] = za;
                saved[sv_at(p, SV_FC0 + FC_B_IN)] = zb;
            }
        }
    }
    __syncthreads();

    // ---- apply from registers, the only write of y
    g.refresh();
#pragma unroll
    for (int r = 0; r < RMAX; ++r) mono_forget(d[r]);
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        {
            const int pn = g.plane(r);
            const float ca = pmu[pn], cb = pm2[pn];
            float ov[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                ov[q] = fmaf(ca, melem<T, VEC>(d[r], q), cb);
                if constexpr (EPI) ov[q] = relu ? fmaxf(ov[q], 0.f) : ov[q];
            }
            if (!TAIL || y) mstore<T, VEC>(g.rsrc(y, c, r), g.off(r), mpack<T, VEC>(ov));
            if constexpr (TAIL) {  // z = relu(bn1(y)) from the same registers
                const float za = pza[pn];
                float oz[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) oz[q] = fmaxf(fmaf(za, melem<T, VEC>(d[r], q), zb), 0.f);
                mstore<T, VEC>(g.rsrc((T*)tl.z, c, r), g.off(r), mpack<T, VEC>(oz));
            }
        }
    }
    }  // (a workgroup past the end of its XCD's range has nothing to do)
}

// ================================================================================================
// backward
// ================================================================================================
// RELOAD (two phases, as cnsn_wide_kernels.h's backward): G and x of 16 slot rows are 64+ data registers, which leaves ONE
// 1024-thread workgroup per CU with its phases in series (16-bit (256,1024,14,14): 0.37-0.39 of the HBM peak).  Here the
// rows are taken in two halves: load a half, mask it, take its sums, drop it; after the algebra load the halves AGAIN
// (read a few microseconds ago with the default cache policy: L2 / Infinity Cache), mask, apply, store.  Half the data
// registers -> 64 VGPRs -> two workgroups per CU whose phases overlap.
// (NH_ = 2: halves; 4: quarters — 16-byte vectors, where a half would be 64 data registers again)
template <typename T, int VEC, int LPP, int RMAX, bool EPI, bool TAIL = false, int NH_ = 1>
__global__ __launch_bounds__(kMonoBlock, NH_ > 1 ? 8 : 4) void mono_bwd_kernel(MonoArgs ma, const T* __restrict__ gy, const T* __restrict__ x,
                                                                  const T* __restrict__ addend, T* __restrict__ dx, GateDev gg,
                                                                  GateDev gf, GateGradDev dgr, GateGradDev dfr,
                                                                  const double* __restrict__ saved, int add, int relu, TailDev tl) {
    constexpr bool RELOAD = NH_ > 1;
    static_assert(!(TAIL && RELOAD), "the tail variant holds the whole channel");
    constexpr int NH = NH_, CNT = RMAX / NH;  // parts of the slot rows, rows per part
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MidArgs a = ma.mid;
    a.sn_two = 0;  // (the host does not send the two-gate form here: the second gate's state would cost ~20 VGPRs)
    const int N = a.N, C = a.C;
    const int npad = kMonoWaves * ma.R * (64 / LPP);
    const MonoWalk wk(C);
    mono_startup_skew();
    if (wk.j < wk.count) {
    const int c = wk.start + wk.j;
    char* lds = smem;
    float* psi = (float*)lds;  // [N] float(mu_c): the shift of the second sum
    float* pfa = psi + npad;    // [N] forward slope  (ReLU mask)   -> later xr
    float* pfb = pfa + npad;    // [N] forward offset (ReLU mask)   -> later c0
    float* ps1 = pfb + npad;    // [N] sum G                         -> later cG
    float* ps2 = ps1 + npad;    // [N] sum G * (x - mu)              -> later cX
    float* pxr = ps2 + npad;
    float* pc0 = pxr + npad;
    float* pb1 = pc0 + (TAIL ? npad : 0);   // [N] TAIL: sum of the masked Gz
    float* pb2 = pb1 + (TAIL ? npad : 0);   // [N] TAIL: sum of the masked Gz * (X - mu)
    double* red = (double*)(pb2 + npad);
    double* psv = red + kMonoWaves * 4;  // [5][N] the five rows of `saved` the algebra needs (parked in LDS: the
                                         // registers belong to the planes until the sums are done)
    const size_t P = (size_t)N * C;
    MonoGeom<T, VEC, LPP> g(ma);
    g.set_channel(c);

    // ---- what the sums and the algebra need from `saved`, a thread per plane, ahead of the bulk loads
    const int n = threadIdx.x;
    const bool act = n < N;
    if (act) {
        const SvRec pme = sv_rec(n, c, N);
        const double mu = saved[sv_at(pme, SV_MU_C)];
        psv[0 * npad + n] = mu;
        psv[1 * npad + n] = saved[sv_at(pme, SV_MU_P)];
        psv[2 * npad + n] = saved[sv_at(pme, SV_SIG_P)];
        psv[3 * npad + n] = saved[sv_at(pme, SV_G)];
        psv[4 * npad + n] = saved[sv_at(pme, SV_ZH_G)];
        psi[n] = (float)mu;
        if ((EPI && relu) || TAIL) {  // (TAIL: the coefficients of z = A * X + B, for ITS ReLU mask)
            pfa[n] = (float)saved[sv_at(pme, SV_FC0 + FC_A_IN)];
            pfb[n] = (float)saved[sv_at(pme, SV_FC0 + FC_B_IN)];
        }
    }
    const float w_g0 = gg.w[2 * c], w_g1 = gg.w[2 * c + 1], gam_g = gg.gamma[c];
    const double rs_g = saved[SV_ROWS * P + c];
    double t_m2 = 0.0, t_r2 = 1.0, t_gam = 0.0;
    if constexpr (TAIL) {
        t_m2 = (double)tl.stats[c];
        t_r2 = (double)tl.stats[C + c];
        t_gam = (double)tl.weight[c];
    }

    // ---- the reads of G and x (+ addend): the whole channel, or one half of its slot rows at a time (RELOAD)
    MRaw<T, VEC> dg_[CNT], dx_[CNT];
    auto load_rows = [&](int h) {
#pragma unroll
        for (int rr = 0; rr < CNT; ++rr) {
            const int r = h * CNT + rr;
            if constexpr (RELOAD) {
                dg_[rr] = mload_cached<T, VEC>(g.rsrc(gy, c, r), g.off(r));
                dx_[rr] = mload_cached<T, VEC>(g.rsrc(x, c, r), g.off(r));
            } else {
                // (TAIL: y may have had no consumer of its own — a NULL gradient is a zero gradient)
                dg_[rr] = mload<T, VEC>(g.rsrc(gy, c, r), (!TAIL || gy) ? g.off(r) : 0x7ffffff8);
                dx_[rr] = mload<T, VEC>(g.rsrc(x, c, r), g.off(r));
            }
        }
        if constexpr (EPI) {
            if (add == ADD_PRE) {
                constexpr int CH = (CNT * VEC * (int)sizeof(T) > 64) ? CNT / 2 : CNT;
#pragma unroll
                for (int r0 = 0; r0 < CNT; r0 += CH) {
                    MRaw<T, VEC> q[CH];
#pragma unroll
                    for (int rr = 0; rr < CH; ++rr) {
                        const int r = h * CNT + r0 + rr;
                        q[rr] = RELOAD ? mload_cached<T, VEC>(g.rsrc(addend, c, r), g.off(r))
                                       : mload<T, VEC>(g.rsrc(addend, c, r), g.off(r));
                    }
#pragma unroll
                    for (int rr = 0; rr < CH; ++rr) dx_[r0 + rr] = madd<T, VEC>(dx_[r0 + rr], q[rr]);
                }
            }
        }
    };
    // ReLU mask of row rr of the loaded rows (forward affine re-evaluated with the coefficients the forward used)
    auto mask_row = [&](int rr, int pi) {
        if constexpr (EPI) {
            if (relu) {
                const float fa = pfa[pi], fb = pfb[pi];
                float gm[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float t = fmaf(fa, melem<T, VEC>(dx_[rr], q) - 0.f, fb);
                    gm[q] = relu_open_r<T>(t) ? melem<T, VEC>(dg_[rr], q) : 0.f;
                }
                dg_[rr] = mpack<T, VEC>(gm);
            }
        }
    };
    load_rows(0);
    __syncthreads();  // psi / pfa / pfb are staged
    g.refresh();
    if constexpr (TAIL) {
        // ---- the gradient of z joins: H = Gy + gamma2 * rstd2 * (Gz where z > 0), z re-evaluated with the forward's own
        //      coefficients; the BatchNorm2d backward's two sums over the masked Gz are taken on the way
        const float kz = (float)(t_gam * t_r2);
        constexpr int CH = (RMAX * VEC * (int)sizeof(T) > 64) ? RMAX / 2 : RMAX;  // bound the transient registers
#pragma unroll
        for (int r0 = 0; r0 < RMAX; r0 += CH) {
            MRaw<T, VEC> q[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r) q[r] = mload<T, VEC>(g.rsrc((const T*)tl.gz, c, r0 + r), g.off(r0 + r));
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int pn = g.plane(r0 + r);
                const float za = pfa[pn], zb = pfb[pn], si = psi[pn];
                float h[VEC], b1 = 0.f, b2 = 0.f;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float X = melem<T, VEC>(dx_[r0 + r], e);
                    const float gz = relu_open_r<T>(fmaf(za, X, zb)) ? melem<T, VEC>(q[r], e) : 0.f;
                    h[e] = fmaf(kz, gz, melem<T, VEC>(dg_[r0 + r], e));
                    b1 += gz;  // (lanes that are not ok() loaded zeros)
                    b2 = fmaf(gz, X - si, b2);
                }
                dg_[r0 + r] = mpack<T, VEC>(h);
                b1 = mono_group_sum<LPP>(b1);
                b2 = mono_group_sum<LPP>(b2);
                if (g.vl == 0 && g.ok(r0 + r)) {
                    pb1[pn] = b1;
                    pb2[pn] = b2;
                }
            }
        }
    }

    // ---- ReLU mask and per-plane sums
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        if (RELOAD && h > 0) {
            load_rows(h);
            g.refresh();
        }
#pragma unroll
        for (int rr = 0; rr < CNT; ++rr) {
            const int r = h * CNT + rr;
            {
                const int pn = g.plane(r);
                const int pi = pn;
                const float si = psi[pi];
                mask_row(rr, pi);
                float s1 = 0.f, s2 = 0.f;
                if (g.ok(r)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = melem<T, VEC>(dg_[rr], q), X = melem<T, VEC>(dx_[rr], q);
                        s1 += G;
                        s2 = fmaf(G, X - si, s2);
                    }
                }
                s1 = mono_group_sum<LPP>(s1);
                s2 = mono_group_sum<LPP>(s2);
                if (g.vl == 0 && g.ok(r)) {
                    ps1[pn] = s1;
                    ps2[pn] = s2;
                }
            }
            if ((rr & (MONO_ILP - 1)) == MONO_ILP - 1) __builtin_amdgcn_sched_barrier(0);  // bound the interleaving (registers)
        }
    }
    __syncthreads();

    // ---- gate / BatchNorm backward, a thread per plane; coefficients of dx
    using Rr = float;
    BwdSumsT<Rr> sums{};
    Rr dtg = 0.f, dtf = 0.f;
    double r_mu = 0, r_mup = 0, r_sigp = 0, r_g = 0, r_zhg = 0;
    const double r_f = 1.0, r_zhf = 0.0;
    if (act) {
        r_mu = psv[0 * npad + n];
        r_mup = psv[1 * npad + n];
        r_sigp = psv[2 * npad + n];
        r_g = psv[3 * npad + n];
        r_zhg = psv[4 * npad + n];
    }
    if (act) sums = fix_sums<Rr>(a, ps1[n], ps2[n], 0.f, 0.f, r_mu, 0.0);
    double t_q1 = 0.0, t_q2 = 0.0;  // TAIL: gamma2*rstd2 * {sum gz', sum gz'*yhat} / (N*M)
    if constexpr (TAIL) {
        // ---- BatchNorm2d backward (wideresnet_cnsn.py:76-77): d bias = sum gz', d weight = sum gz' * yhat with
        //      yhat = rstd2 * (g * X - mean2); its dependence on the batch statistics changes the gradient reaching y by
        //      -q1 - q2 * yhat(X): fold that into the plane sums SelfNorm's backward works from
        double tt[2] = {0.0, 0.0};
        if (act) {
            const double b1 = pb1[n], b2 = (double)pb2[n] + ((double)(float)r_mu - r_mu) * (double)pb1[n];
            tt[0] = b1;
            tt[1] = r_g * (b2 + r_mu * b1) - t_m2 * b1;
        }
        mono_block_sum<2>(tt, red);
        const double T1 = tt[0], T2 = t_r2 * tt[1];
        if (threadIdx.x == 0) {
            tl.d_weight[c] = (float)T2;
            tl.d_bias[c] = (float)T1;
        }
        if (tl.training) {
            const double inv = a.inv_n / (double)a.M;
            t_q1 = t_gam * t_r2 * T1 * inv;
            t_q2 = t_gam * t_r2 * T2 * inv;
        }
        if (act) {
            const double M_ = (double)a.M, M2n = (r_sigp * r_sigp - (double)a.eps_sn) * (M_ - 1.0);
            sums.S1in = (Rr)((double)sums.S1in - t_q2 * t_r2 * r_g * M_ * r_mu + M_ * (t_q2 * t_r2 * t_m2 - t_q1));
            sums.S2in = (Rr)((double)sums.S2in - t_q2 * t_r2 * r_g * M2n);
        }
    }
    if (act) gate_dt<Rr>(a, sums, Rr(1), (Rr)r_mu, Rr(0), (Rr)r_mup, (Rr)r_g, (Rr)r_f, dtg, dtf);
    double s4[2] = {(double)dtg, (double)dtg * r_zhg};
    mono_block_sum<2>(s4, red);
    BnBwd b{};
    b.s_dt_g = s4[0];
    b.s_dtz_g = s4[1];
    b.wg0 = w_g0;
    b.wg1 = w_g1;
    b.kg = (double)gam_g * rs_g;
    double sw[2] = {0, 0};
    if (act) {
        const BwdPlaneT<Rr> o = bwd_plane<Rr>(a, b, sums, (double)dtg, (double)dtf, r_zhg, r_zhf, (Rr)r_g, (Rr)r_f, Rr(1), Rr(1),
                                              (Rr)r_mu, (Rr)r_mup, (Rr)r_sigp, Rr(1), Rr(0));
        sw[0] = (double)o.dz_g * r_mup;
        sw[1] = (double)o.dz_g * r_sigp;
        BwdCoefs k = bwd_coefs<Rr>(a, o, Rr(0), Rr(0), (Rr)r_g, Rr(1), (Rr)r_mu, (Rr)r_mup, r_mu, Rr(1), r_mu, Rr(1));
        if constexpr (TAIL) {  // dx = g * (H - q1 - q2 * yhat(X)) + ...: the X term joins the slope, the rest the constant
            const double e = r_g * r_g * t_q2 * t_r2;
            k.c0_in = (float)((double)k.c0_in - e * (double)k.xr_in + r_g * (t_q2 * t_r2 * t_m2 - t_q1));
            k.cX_in = (float)((double)k.cX_in - e);
        }
        ps1[n] = k.cG_in;  // (every thread read its own ps1 / ps2 before the reduction above)
        ps2[n] = k.cX_in;
        pxr[n] = k.xr_in;
        pc0[n] = k.c0_in;
    }
    mono_block_sum<2>(sw, red);
    if (threadIdx.x == 0) {
        dgr.dgamma[c] = (float)s4[1];
        dgr.dbeta[c] = (float)s4[0];
        dgr.dw[2 * c] = (float)sw[0];
        dgr.dw[2 * c + 1] = (float)sw[1];
    }
    // (mono_block_sum ends with every thread past its second barrier: the coefficient rows are visible)

    // ---- dx, the only write: from the registers, or (RELOAD) from the halves read again
    g.refresh();
    if constexpr (!RELOAD) {
#pragma unroll
        for (int r = 0; r < CNT; ++r) {
            mono_forget(dg_[r]);
            mono_forget(dx_[r]);
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        if constexpr (RELOAD) {
            load_rows(h);
            g.refresh();
        }
#pragma unroll
        for (int rr = 0; rr < CNT; ++rr) {
            const int r = h * CNT + rr;
            {
                const int pn = g.plane(r);
                const int pi = pn;
                if constexpr (RELOAD) mask_row(rr, pi);
                const float cG = ps1[pi], cX = ps2[pi], xr = pxr[pi], c0 = pc0[pi];
                float ov[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    ov[q] = fmaf(cG, melem<T, VEC>(dg_[rr], q), fmaf(cX, melem<T, VEC>(dx_[rr], q) - xr, c0));
                mstore<T, VEC>(g.rsrc(dx, c, r), g.off(r), mpack<T, VEC>(ov));
            }
            if ((rr & (MONO_ILP - 1)) == MONO_ILP - 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    }  // (a workgroup past the end of its XCD's range has nothing to do)
}

}  // namespace cnsn
