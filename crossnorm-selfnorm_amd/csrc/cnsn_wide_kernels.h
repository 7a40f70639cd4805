// Channel-GROUP-in-registers strategy ("wide") for planes that are NOT a whole number of 8- or 16-byte vectors — 7x7:
// 98 bytes in 16 bits, 196 in fp32 (ResNet-50 stage 4, WideResNet's last sites are 8x8 and do not need it).
//
// The mono kernels reach such planes one element per lane (2- or 4-byte accesses, 49 of 64 lanes) and the channel-local
// kernels through LDS: (256,2048,7,7) bf16 ran at 27 % of its HBM bound.  Here ONE 1024-thread workgroup takes CH
// ADJACENT channels: the CH planes of an instance are contiguous in NCHW — CH*M elements — and CH = VEC (the elements
// of one 16- or 8-byte vector) makes that super-plane exactly M vectors: lane l < M holds elements l*VEC .. l*VEC+VEC-1,
// which belong to channel (l*VEC)/M and, past a per-lane split point, to the next one.  A wave holds R instances
// (16 waves: N <= 16*R), so every access is a full 16-byte vector: 8 channels per workgroup in 16 bits, 4 in fp32.
// Forward: the whole channel group sits in registers — one launch, single touch, no exchange.  Backward: G and x of a
// group do not fit the registers of one workgroup (128 VGPRs per lane), so it reads them twice, the second time from
// cache (see wide_bwd_kernel).
//   per-(instance, channel) sums: every lane forms the two partial sums of its two channels, the partials of a batch of
//   rows go to a per-wave LDS scratch, and one lane per (row, channel) adds the 7-14 partials of its segment in a fixed
//   order (no atomics: results are reproducible); exact two-pass statistics as everywhere else;
//   SelfNorm's BatchNorm1d over N: one thread per (instance, channel), channel = thread index mod CH, so a channel's sum
//   is a butterfly over lanes CH apart + 16 wave partials.
// Algebra and `saved` contract are the mono / channel-local kernels' (SelfNorm alone, one gate, optional PRE add and
// ReLU), so a forward of this strategy can be followed by a backward of any other.
#pragma once
#include "cnsn_mono_kernels.h"

namespace cnsn {

constexpr int kWideBlock = 1024, kWideWaves = 16, kWideRows = 16;

struct WideArgs {
    MidArgs mid;
    int R;  // instances per wave (<= 16): N <= 16 * R
};

// LDS: scratch of one batch of rows [16 waves][BATCH][64 lanes][NACC pairs], then NARR arrays over the (instance,
// channel) pairs, then the reduction scratch [16][8][4] doubles
// (the scratch doubles as the place where `parked_bytes` per thread of plane rows wait during the algebra)
// (perm_ints: the CrossNorm variants keep the batch permutation / its inverse there, N ints)
__host__ __device__ inline size_t wide_lds_bytes(int N, int ch, int batch, int nacc, int narr, int parked_bytes,
                                                 int perm_ints = 0) {
    const size_t scratch = (size_t)kWideWaves * batch * 64 * nacc * 8, park = (size_t)kWideBlock * parked_bytes;
    return (scratch > park ? scratch : park) + (size_t)narr * (((size_t)N * ch + 63) & ~(size_t)63) * 4 +
           (((size_t)perm_ints * 4 + 63) & ~(size_t)63) + (size_t)kWideWaves * 8 * 4 * 8;
}
constexpr int kWideParkFwd = 8;  // rows of x parked during the forward's algebra

// per-lane geometry of the super-plane
template <int VEC>
struct WideLane {
    int a;     // channel (within the group) of this lane's first element
    int qs;    // elements 0 .. qs-1 belong to channel a, the rest to a + 1
    bool live;  // lane < M
    __device__ __forceinline__ WideLane(int lane, int M) {
        a = (lane * VEC) / M;
        const int left = (a + 1) * M - lane * VEC;
        qs = left < VEC ? left : VEC;
        live = lane < M;
        if (!live) {
            a = 0;
            qs = VEC;
        }
    }
};

// sum of v over the threads of the workgroup whose index is congruent to this thread's modulo CH (CH a power of two
// <= 8): butterfly over lanes CH, 2CH, .. 32 apart, then the 16 waves' partials through `red` ([16][8][NACC] doubles)
template <int NACC, int CH>
__device__ __forceinline__ void wide_chan_sum(double (&v)[NACC], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = CH; m < 64; m <<= 1)
#pragma unroll
        for (int i = 0; i < NACC; ++i) v[i] += __shfl_xor(v[i], m);
    __syncthreads();  // (the previous use of `red` is over)
    if (lane < CH) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) red[(wave * 8 + lane) * NACC + i] = v[i];
    }
    __syncthreads();
    const int k = lane & (CH - 1);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kWideWaves; ++w) s += red[(w * 8 + k) * NACC + i];
        v[i] = s;
    }
}

// One batch of rows: every lane has written its NACC (pa, pb) pairs per row to sc[(wave*BATCH + rr)*64 + lane][i];
// lane p < BATCH*CH takes (rr, k) = (p / CH, p % CH) and returns the segment sums of channel k in row rr.
template <int VEC, int NACC, int BATCH>
__device__ __forceinline__ void wide_segment(const float2* sc, int wave, int p, int M, float (&out)[NACC]) {
    constexpr int CH = VEC;
    const int rr = p / CH, k = p - rr * CH;
    const int lo = (k * M) / VEC, hi = (k * M + M - 1) / VEC;
#pragma unroll
    for (int i = 0; i < NACC; ++i) out[i] = 0.f;
    const float2* row = sc + ((size_t)(wave * BATCH + rr) * 64) * NACC;
    for (int l = lo; l <= hi; ++l) {
        const bool first = (l * VEC) / M == k;  // the lane's first channel is k: its pa; otherwise (only l == lo) its pb
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const float2 v = row[l * NACC + i];
            out[i] += first ? v.x : v.y;
        }
    }
}

// ================================================================================================
// forward
// ================================================================================================
// CN: CrossNorm without crop boxes ahead of SelfNorm (cn_op_2ins_space_chan, models/cnsn.py:58-91 with crop 'neither'): the
// batch permutation pairs planes of ONE channel, and the whole channel group is in this workgroup — a plane's style
// statistics are read from the LDS arrays at index perm[n] (same scheme as cnsn_mono_cn_kernels.h).
template <typename T, int VEC, bool EPI, bool CN = false>
__global__ __launch_bounds__(kWideBlock) void wide_fwd_kernel(WideArgs wa, const T* __restrict__ x, const T* __restrict__ addend,
                                                              T* __restrict__ y, GateDev gg, double* __restrict__ saved,
                                                              int add, int relu, const int64_t* __restrict__ perm) {
    constexpr int CH = VEC, VB = VEC * (int)sizeof(T), BATCH = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = wa.mid;
    const int N = a.N, C = a.C, M = a.M, R = wa.R;
    const int npad = (N * CH + 63) & ~63;
    float2* sc = (float2*)smem;                                              // [16][BATCH][64]
    constexpr size_t kScratch = (size_t)kWideWaves * BATCH * 64 * 8, kPark = (size_t)kWideBlock * kWideParkFwd * VB;
    float* smu = (float*)(smem + (kScratch > kPark ? kScratch : kPark));     // [N][CH] mean   -> later a_in
    float* sm2 = smu + npad;                                                 // [N][CH] M2     -> later b_in
    // CN: one more coefficient (xr), and the statistics have to stay readable until EVERY pair has formed its coefficients
    // (another pair reads them as its style source): the coefficients wait in registers across a barrier before they
    // replace the statistics in place
    float* sca = smu;                                                        // [N][CH] a_in
    float* scb = sm2;                                                        // [N][CH] b_in
    float* sxr = CN ? sm2 + npad : nullptr;                                  // [N][CH] xr
    int* sperm = CN ? (int*)(sxr + npad) : nullptr;                          // [N] style source of every instance
    double* red = (double*)(CN ? (char*)sperm + (((size_t)N * 4 + 63) & ~(size_t)63) : (char*)(sm2 + npad));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const MonoWalk wk(C / CH);  // neighbouring channel groups (they share 128-byte lines) on one XCD
    if (wk.j >= wk.count) return;  // (workgroup-uniform: the grid is rounded up to whole rounds of the 8 XCDs)
    const int c0 = (wk.start + wk.j) * CH;
    if constexpr (CN) {
        for (int n = threadIdx.x; n < N; n += kWideBlock) sperm[n] = (int)perm[n];  // (read behind the statistics' barriers)
    }
    const size_t P = (size_t)N * C;
    const WideLane<VEC> wl(lane, M);
    const int voff = lane * VB;
    const int gbytes = CH * M * (int)sizeof(T);  // a super-plane

    // ---- the only read of x (+ addend)
    MRaw<T, VEC> d[kWideRows];
#pragma unroll
    for (int r = 0; r < kWideRows; ++r) {
        const int n = wave * R + r;
        const bool ok = r < R && n < N;  // wave-uniform
        const size_t off = ((size_t)(ok ? n : 0) * C + c0) * M;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + off), 0, ok ? gbytes : 0, 0x00020000);
        d[r] = mload<T, VEC>(rs, voff);
        if constexpr (EPI) {
            if (add == ADD_PRE) {
                const __amdgpu_buffer_rsrc_t ra =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(addend + off), 0, ok ? gbytes : 0, 0x00020000);
                d[r] = madd<T, VEC>(d[r], mload<T, VEC>(ra, voff));
            }
        }
    }

    // ---- exact two-pass statistics of every (instance, channel) plane
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int h = 0; h < kWideRows / BATCH; ++h) {
            if (h * BATCH >= R) break;  // (workgroup-uniform: no rows in this batch)
#pragma unroll
            for (int rr = 0; rr < BATCH; ++rr) {
                const int r = h * BATCH + rr;
                const int n = wave * R + r;
                float ma = 0.f, mb = 0.f;
                if (pass == 1 && r < R && n < N) {
                    ma = smu[n * CH + wl.a];
                    mb = smu[n * CH + (wl.a + 1 < CH ? wl.a + 1 : wl.a)];
                }
                float pa = 0.f, pb = 0.f;
                mono_forget(d[r]);  // (keeps the unpacked floats of other rows / passes out of the registers)
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float f = melem<T, VEC>(d[r], q);  // dead lanes and dead rows loaded zeros
                    const bool first = q < wl.qs;
                    const float t = pass == 0 ? f : (first ? f - ma : f - mb);
                    const float u = pass == 0 ? t : t * t;
                    pa += (first && wl.live) ? u : 0.f;
                    pb += (!first && wl.live) ? u : 0.f;
                }
                sc[(size_t)(wave * BATCH + rr) * 64 + lane] = make_float2(pa, pb);
                if (rr & 1) __builtin_amdgcn_sched_barrier(0);  // bound the interleaving of rows (registers)
            }
            __syncthreads();
            for (int p = lane; p < BATCH * CH; p += 64) {
                float s[1];
                wide_segment<VEC, 1, BATCH>(sc, wave, p, M, s);
                const int r = h * BATCH + p / CH, k = p % CH;
                const int n = wave * R + r;
                if (r < R && n < N) {
                    if (pass == 0)
                        smu[n * CH + k] = s[0] / (float)M;
                    else
                        sm2[n * CH + k] = s[0];
                }
            }
            __syncthreads();
        }
    }

    // ---- the algebra wants registers: the last rows of the planes wait in the (now idle) scratch
    constexpr int NPARKED = kWideParkFwd;
    MRaw<T, VEC>* parked = (MRaw<T, VEC>*)sc;
#pragma unroll
    for (int i = 0; i < NPARKED; ++i) parked[(size_t)i * kWideBlock + threadIdx.x] = d[kWideRows - NPARKED + i];

    // ---- gates: BatchNorm1d over the N planes of each channel, a thread per (instance, channel)  (cnsn.py:137-141)
    using Rr = float;
    constexpr int PP = 2;  // pairs per thread: N * CH <= 2048
    const int k = threadIdx.x & (CH - 1), c = c0 + k;
    // (CrossNorm ALONE — sn_active 0, CN instantiations only: no gate, g = 1, nothing of `gg` is touched; models/cnsn.py:152-164
    //  with selfnorm=None)
    const bool sn = !CN || a.sn_active != 0;
    const double wg0 = sn ? gg.w[2 * c] : 0.f, wg1 = sn ? gg.w[2 * c + 1] : 0.f, gam = sn ? gg.gamma[c] : 0.f, bet = sn ? gg.beta[c] : 0.f;
    const double rm0 = sn ? gg.run_mean[c] : 0.f, rv0 = sn ? gg.run_var[c] : 1.f;
    auto plane_of = [&](int p) {  // (cheap: recomputed after the reduction instead of kept across it)
        MomentsT<Rr> o;
        o.mu_c = o.mu_s = smu[p];
        o.M2c = o.M2s = sm2[p];
        o.mu_o = o.M2o = 0.f;
        if constexpr (CN) {
            const int src = sperm[p / CH] * CH + (p & (CH - 1));  // the same channel of the style instance (cnsn.py:62,66)
            return fwd_plane<Rr>(a, o, smu[src], sm2[src]);
        } else {
            return fwd_plane<Rr>(a, o, 0.f, 0.f);
        }
    };
    double sz[2] = {0.0, 0.0};
#pragma unroll 1  // (one pair's algebra at a time: the planes keep their registers)
    for (int i = 0; i < PP; ++i) {
        const int p = threadIdx.x + i * kWideBlock;
        if (p / CH < N) {
            const FwdPlaneT<Rr> f = plane_of(p);
            const double z = wg0 * (double)f.mu_p + wg1 * (double)f.sig_p;
            sz[0] += z;
            sz[1] += z * z;
        }
    }
    double mg = rm0, rg;
    if (!sn) {
        rg = 1.0;
    } else if (a.sn_training) {
        wide_chan_sum<2, CH>(sz, red);
        mg = sz[0] * a.inv_n;
        double vg = sz[1] * a.inv_n - mg * mg;
        vg = vg > 0.0 ? vg : 0.0;
        rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
        if (threadIdx.x < CH) {
            const double mom_ = a.momentum, unb = a.unbias_n;
            gg.run_mean[c] = (float)((1.0 - mom_) * rm0 + mom_ * mg);
            gg.run_var[c] = (float)((1.0 - mom_) * rv0 + mom_ * vg * unb);
            if (c == 0) bump_batches_tracked(gg.nbt);
        }
    } else {
        rg = (double)__builtin_amdgcn_rsqf((float)rv0 + a.eps_bn);
    }
    if (saved && sn && threadIdx.x < CH) {
        saved[SV_ROWS * P + c] = rg;
        saved[SV_ROWS * P + C + c] = 1.0;
    }
    float ka[PP], kb[PP], kx[PP];  // CN: this thread's coefficients, written behind the barrier below
#pragma unroll 1
    for (int i = 0; i < PP; ++i) {
        const int p = threadIdx.x + i * kWideBlock;
        ka[i] = kb[i] = kx[i] = 0.f;
        if (p / CH < N) {
            const int n = p / CH;
            const FwdPlaneT<Rr> f = plane_of(p);
            const double z = wg0 * (double)f.mu_p + wg1 * (double)f.sig_p;
            const double zhg = sn ? (z - mg) * rg : 0.0;
            const Rr gt = sn ? sigmoid_r<Rr>((Rr)(gam * zhg + bet)) : Rr(1);
            const FwdCoefs cf = fwd_coefs<Rr>(a, f, gt, 1.f);
            if (saved) {
                const SvRec ps = sv_rec(n, c, N);
                store_fwd_plane<Rr>(saved, P, ps, f, CN ? 1 : 0);
                saved[sv_at(ps, SV_G)] = gt;
                saved[sv_at(ps, SV_ZH_G)] = zhg;
                saved[sv_at(ps, SV_F)] = 1.0;
                saved[sv_at(ps, SV_ZH_F)] = 0.0;
                if (a.save_coefs) store_fwd_coefs(saved, ps, cf);
            }
            if constexpr (CN) {
                ka[i] = cf.a_in;
                kb[i] = cf.b_in;
                kx[i] = cf.xr;
            } else {
                sca[p] = cf.a_in;  // SelfNorm alone: y = a_in * x + b_in (xr = 0); only this thread reads smu[p] / sm2[p]
                scb[p] = cf.b_in;
            }
        }
    }
    if constexpr (CN) {
        __syncthreads();  // every pair has read the statistics of its style source
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int p = threadIdx.x + i * kWideBlock;
            if (p / CH < N) {
                sca[p] = ka[i];  // y = a_in * (x - xr) + b_in
                scb[p] = kb[i];
                sxr[p] = kx[i];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPARKED; ++i) d[kWideRows - NPARKED + i] = parked[(size_t)i * kWideBlock + threadIdx.x];

    // ---- apply from registers, the only write of y
#pragma unroll
    for (int r = 0; r < kWideRows; ++r) {
        const int n = wave * R + r;
        const bool ok = r < R && n < N;
        if (!ok) continue;  // wave-uniform
        const int ia = n * CH + wl.a, ib = n * CH + (wl.a + 1 < CH ? wl.a + 1 : wl.a);
        const float ca = sca[ia], cb = scb[ia], ca2 = sca[ib], cb2 = scb[ib];
        const float xa = CN ? sxr[ia] : 0.f, xb = CN ? sxr[ib] : 0.f;
        mono_forget(d[r]);
        float ov[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const bool first = q < wl.qs;
            if constexpr (CN)
                ov[q] = fmaf(first ? ca : ca2, melem<T, VEC>(d[r], q) - (first ? xa : xb), first ? cb : cb2);
            else
                ov[q] = fmaf(first ? ca : ca2, melem<T, VEC>(d[r], q), first ? cb : cb2);
            if constexpr (EPI) ov[q] = relu ? fmaxf(ov[q], 0.f) : ov[q];
        }
        const size_t off = ((size_t)n * C + c0) * M;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(y + off), 0, gbytes, 0x00020000);
        mstore<T, VEC>(rs, voff, mpack<T, VEC>(ov));
        if (r & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// ================================================================================================
// backward, two-phase ("reload"): 16-byte vectors = as many channels per workgroup as the forward
// ================================================================================================
// Holding G and x of a group at once would limit the backward to 8-byte vectors = half the channels per workgroup =
// TWO rounds of the chip (built first: 95 us at (256,2048,7,7) bf16 against 60 us for this one).  This kernel walks the
// group's instances in two halves of 8 rows per wave: phase 1
// loads a half (G and x: 64 registers), takes its sums and drops it; after the algebra, phase 2 loads the halves AGAIN
// (the group's 2 x 200 KB were just read: L2 / Infinity Cache), applies and stores.  One round of the chip, every access
// 16 bytes; G and x are read twice, of which once from cache.
template <typename T, int VEC, bool EPI, bool CN = false>
__global__ __launch_bounds__(kWideBlock) void wide_bwd_kernel(WideArgs wa, const T* __restrict__ gy, const T* __restrict__ x,
                                                               const T* __restrict__ addend, T* __restrict__ dx, GateDev gg,
                                                               GateGradDev dgr, const double* __restrict__ saved, int add,
                                                               int relu, const int64_t* __restrict__ perm) {
    constexpr int CH = VEC, VB = VEC * (int)sizeof(T), BATCH = 4, HALF = 8, PP = 2;
    static_assert(VB == 16, "full vectors");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MidArgs a = wa.mid;
    a.sn_two = 0;
    const int N = a.N, C = a.C, M = a.M, R = wa.R;
    const int npad = (N * CH + 63) & ~63;
    float2* sc = (float2*)smem;                                               // [16][BATCH][64][2]
    float* psi = (float*)(smem + (size_t)kWideWaves * BATCH * 64 * 2 * 8);    // [N][CH] float(mu_c): shift of the second sum
    float* pfa = psi + npad;                                                  // forward slope  (ReLU mask)
    float* pfb = pfa + npad;                                                  // forward offset (ReLU mask)
    float* ps1 = pfb + npad;                                                  // sum G            -> later cG
    float* ps2 = ps1 + npad;                                                  // sum G*(x - mu)   -> later cX
    float* pxr = ps2 + npad;
    float* pc0 = pxr + npad;
    // CN: the forward's xr (ReLU mask), what a pair sends to its style source (Emu, Esig), the inverse permutation
    float* pfx = CN ? pc0 + npad : nullptr;
    float* pem = CN ? pfx + npad : nullptr;
    float* pes = CN ? pem + npad : nullptr;
    int* iperm = CN ? (int*)(pes + npad) : nullptr;
    double* red = (double*)(CN ? (char*)iperm + (((size_t)N * 4 + 63) & ~(size_t)63) : (char*)(pc0 + npad));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const MonoWalk wk(C / CH);
    if (wk.j >= wk.count) return;
    const int c0 = (wk.start + wk.j) * CH;
    if constexpr (CN) {
        for (int n = threadIdx.x; n < N; n += kWideBlock) iperm[(int)perm[n]] = n;  // instance perm[n] lent its statistics to n
    }
    const size_t P = (size_t)N * C;
    const WideLane<VEC> wl(lane, M);
    const int voff = lane * VB;
    const int gbytes = CH * M * (int)sizeof(T);
    const int k = threadIdx.x & (CH - 1), c = c0 + k;

    // ---- shifts (and the forward's coefficients for the ReLU mask), a thread per (instance, channel)
#pragma unroll 1
    for (int i = 0; i < PP; ++i) {
        const int p = threadIdx.x + i * kWideBlock, np = p / CH;
        if (np < N) {
            const SvRec ps = sv_rec(np, c, N);
            psi[p] = (float)saved[sv_at(ps, SV_MU_C)];
            if (EPI && relu) {
                pfa[p] = (float)saved[sv_at(ps, SV_FC0 + FC_A_IN)];
                pfb[p] = (float)saved[sv_at(ps, SV_FC0 + FC_B_IN)];
                if constexpr (CN) pfx[p] = (float)saved[sv_at(ps, SV_FC0 + FC_XR)];
            }
        }
    }
    const bool sn = !CN || a.sn_active != 0;  // (CrossNorm alone: no gate, no gate gradients — see the forward)
    const float w_g0 = sn ? gg.w[2 * c] : 0.f, w_g1 = sn ? gg.w[2 * c + 1] : 0.f, gam_g = sn ? gg.gamma[c] : 0.f;
    const double rs_g = sn ? saved[SV_ROWS * P + c] : 0.0;
    __syncthreads();

    MRaw<T, VEC> dg_[HALF], dx_[HALF];
    // rows h*HALF .. h*HALF+7 of this wave; `again`: the second read (the lines are expected in cache: no nt hint)
    auto load_half = [&](int h) {
#pragma unroll
        for (int rr = 0; rr < HALF; ++rr) {
            const int r = h * HALF + rr, n = wave * R + r;
            const bool ok = r < R && n < N;
            const size_t off = ((size_t)(ok ? n : 0) * C + c0) * M;
            const int bytes = ok ? gbytes : 0;
            dg_[rr] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(gy + off), 0, bytes, 0x00020000), voff, 0, 0);
            dx_[rr] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(x + off), 0, bytes, 0x00020000), voff, 0, 0);
            if constexpr (EPI) {
                if (add == ADD_PRE)
                    dx_[rr] = madd<T, VEC>(dx_[rr], __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(addend + off), 0, bytes, 0x00020000), voff, 0, 0));
            }
        }
    };
    // ReLU mask of row rr of the loaded half (forward affine re-evaluated with the coefficients the forward used)
    auto mask_row = [&](int rr, int ia, int ib) {
        if constexpr (EPI) {
            if (relu) {
                const float fa = pfa[ia], fb = pfb[ia], fa2 = pfa[ib], fb2 = pfb[ib];
                const float fx = CN ? pfx[ia] : 0.f, fx2 = CN ? pfx[ib] : 0.f;
                float gm[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const bool first = q < wl.qs;
                    const float t = CN ? fmaf(first ? fa : fa2, melem<T, VEC>(dx_[rr], q) - (first ? fx : fx2), first ? fb : fb2)
                                       : fmaf(first ? fa : fa2, melem<T, VEC>(dx_[rr], q), first ? fb : fb2);
                    gm[q] = relu_open_r<T>(t) ? melem<T, VEC>(dg_[rr], q) : 0.f;
                }
                dg_[rr] = mpack<T, VEC>(gm);
            }
        }
    };

    // ---- phase 1: per-plane sums, half by half
#pragma unroll 1
    for (int h = 0; h < kWideRows / HALF; ++h) {
        if (h * HALF >= R) break;
        load_half(h);
#pragma unroll
        for (int bb = 0; bb < HALF / BATCH; ++bb) {
#pragma unroll
            for (int rb = 0; rb < BATCH; ++rb) {
                const int rr = bb * BATCH + rb, r = h * HALF + rr, n = wave * R + r;
                const bool ok = r < R && n < N;
                const int ia = (ok ? n : 0) * CH + wl.a, ib = (ok ? n : 0) * CH + (wl.a + 1 < CH ? wl.a + 1 : wl.a);
                const float sia = psi[ia], sib = psi[ib];
                mono_forget(dg_[rr]);
                mono_forget(dx_[rr]);
                mask_row(rr, ia, ib);
                float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const bool first = q < wl.qs;
                    const float G = melem<T, VEC>(dg_[rr], q), X = melem<T, VEC>(dx_[rr], q);
                    const float t = G * (X - (first ? sia : sib));
                    const bool fa_ = first && wl.live && ok, fb_ = !first && wl.live && ok;
                    a1 += fa_ ? G : 0.f;
                    b1 += fb_ ? G : 0.f;
                    a2 += fa_ ? t : 0.f;
                    b2 += fb_ ? t : 0.f;
                }
                sc[((size_t)(wave * BATCH + rb) * 64 + lane) * 2 + 0] = make_float2(a1, b1);
                sc[((size_t)(wave * BATCH + rb) * 64 + lane) * 2 + 1] = make_float2(a2, b2);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            for (int q = lane; q < BATCH * CH; q += 64) {
                float s[2];
                wide_segment<VEC, 2, BATCH>(sc, wave, q, M, s);
                const int r = h * HALF + bb * BATCH + q / CH, kk = q % CH;
                const int n = wave * R + r;
                if (r < R && n < N) {
                    ps1[n * CH + kk] = s[0];
                    ps2[n * CH + kk] = s[1];
                }
            }
            __syncthreads();
        }
    }

    // ---- gate / BatchNorm backward, a thread per (instance, channel) pair (two pairs per thread, one at a time);
    //      the pair's `saved` rows are read here (and once more below) instead of being held across phase 1
    using Rr = float;
    auto pair_state = [&](int p, BwdSumsT<Rr>& sums, Rr& dtg, Rr& dtf, double& r_mu, double& r_mup, double& r_sigp, double& r_g,
                          double& r_zhg, CnRowsT<Rr>& cr) {
        const SvRec ps = sv_rec(p / CH, c, N);
        r_mu = saved[sv_at(ps, SV_MU_C)];
        r_mup = saved[sv_at(ps, SV_MU_P)];
        r_sigp = saved[sv_at(ps, SV_SIG_P)];
        r_g = saved[sv_at(ps, SV_G)];
        r_zhg = saved[sv_at(ps, SV_ZH_G)];
        if constexpr (CN) {
            cr = load_cn_rows<Rr>(a, saved, ps, r_mu);
            sums = fix_sums<Rr>(a, ps1[p], ps2[p], 0.f, 0.f, r_mu, (double)cr.mu_o);
            if (sn) gate_dt<Rr>(a, sums, cr.a1, cr.m_in, cr.mu_o, (Rr)r_mup, (Rr)r_g, Rr(1), dtg, dtf);
        } else {
            sums = fix_sums<Rr>(a, ps1[p], ps2[p], 0.f, 0.f, r_mu, 0.0);
            gate_dt<Rr>(a, sums, Rr(1), (Rr)r_mu, Rr(0), (Rr)r_mup, (Rr)r_g, Rr(1), dtg, dtf);
        }
    };
    auto plane_bwd = [&](const BnBwd& b, const BwdSumsT<Rr>& sums, Rr dtg, Rr dtf, double r_mu, double r_mup, double r_sigp,
                         double r_g, double r_zhg, const CnRowsT<Rr>& cr) {
        if constexpr (CN)
            return bwd_plane<Rr>(a, b, sums, (double)dtg, 0.0, r_zhg, 0.0, (Rr)r_g, Rr(1), cr.aa, cr.a1, cr.m_in, (Rr)r_mup,
                                 (Rr)r_sigp, cr.sig_c, cr.M2c);
        else
            return bwd_plane<Rr>(a, b, sums, (double)dtg, (double)dtf, r_zhg, 0.0, (Rr)r_g, Rr(1), Rr(1), Rr(1), (Rr)r_mu,
                                 (Rr)r_mup, (Rr)r_sigp, Rr(1), Rr(0));
    };
    double s4[2] = {0.0, 0.0};
#pragma unroll 1
    for (int i = 0; i < PP; ++i) {
        const int p = threadIdx.x + i * kWideBlock;
        if (p / CH < N) {
            BwdSumsT<Rr> sums{};
            CnRowsT<Rr> cr{};
            Rr dtg = 0.f, dtf = 0.f;
            double r_mu, r_mup, r_sigp, r_g, r_zhg;
            pair_state(p, sums, dtg, dtf, r_mu, r_mup, r_sigp, r_g, r_zhg, cr);
            s4[0] += (double)dtg;
            s4[1] += (double)dtg * r_zhg;
        }
    }
    wide_chan_sum<2, CH>(s4, red);
    BnBwd b{};
    b.s_dt_g = s4[0];
    b.s_dtz_g = s4[1];
    b.wg0 = w_g0;
    b.wg1 = w_g1;
    b.kg = (double)gam_g * rs_g;
    double sw[2] = {0, 0};
#pragma unroll 1
    for (int i = 0; i < PP; ++i) {
        const int p = threadIdx.x + i * kWideBlock;
        if (p / CH < N) {
            BwdSumsT<Rr> sums{};
            CnRowsT<Rr> cr{};
            Rr dtg = 0.f, dtf = 0.f;
            double r_mu, r_mup, r_sigp, r_g, r_zhg;
            pair_state(p, sums, dtg, dtf, r_mu, r_mup, r_sigp, r_g, r_zhg, cr);
            const BwdPlaneT<Rr> o = plane_bwd(b, sums, dtg, dtf, r_mu, r_mup, r_sigp, r_g, r_zhg, cr);
            sw[0] += (double)o.dz_g * r_mup;
            sw[1] += (double)o.dz_g * r_sigp;
            if constexpr (CN) {  // what this pair sends to its style source; the coefficients follow behind the barrier
                pem[p] = o.Emu;
                pes[p] = o.Esig;
            } else {
                const BwdCoefs kf = bwd_coefs<Rr>(a, o, Rr(0), Rr(0), (Rr)r_g, Rr(1), (Rr)r_mu, (Rr)r_mup, r_mu, Rr(1), r_mu, Rr(1));
                ps1[p] = kf.cG_in;  // (this thread is the only reader of ps1[p] / ps2[p] as sums)
                ps2[p] = kf.cX_in;
                pxr[p] = kf.xr_in;
                pc0[p] = kf.c0_in;
            }
        }
    }
    wide_chan_sum<2, CH>(sw, red);  // (also the barrier that makes Emu / Esig of every pair visible)
    if (sn && threadIdx.x < CH) {
        dgr.dgamma[c] = (float)s4[1];
        dgr.dbeta[c] = (float)s4[0];
        dgr.dw[2 * c] = (float)sw[0];
        dgr.dw[2 * c + 1] = (float)sw[1];
    }
    if constexpr (CN) {  // dx coefficients: the pair's own terms plus what the instance that borrowed its statistics sent
#pragma unroll 1
        for (int i = 0; i < PP; ++i) {
            const int p = threadIdx.x + i * kWideBlock;
            if (p / CH < N) {
                BwdSumsT<Rr> sums{};
                CnRowsT<Rr> cr{};
                Rr dtg = 0.f, dtf = 0.f;
                double r_mu, r_mup, r_sigp, r_g, r_zhg;
                pair_state(p, sums, dtg, dtf, r_mu, r_mup, r_sigp, r_g, r_zhg, cr);
                const BwdPlaneT<Rr> o = plane_bwd(b, sums, dtg, dtf, r_mu, r_mup, r_sigp, r_g, r_zhg, cr);
                const int src = iperm[p / CH] * CH + (p & (CH - 1));
                const BwdCoefs kf = bwd_coefs<Rr>(a, o, pem[src], pes[src], (Rr)r_g, cr.a1, cr.m_in, (Rr)r_mup, r_mu, cr.sig_c,
                                                  cr.mu_s, cr.sig_s);
                ps1[p] = kf.cG_in;  // (this thread is the only reader of ps1[p] / ps2[p] as sums)
                ps2[p] = kf.cX_in;
                pxr[p] = kf.xr_in;
                pc0[p] = kf.c0_in;
            }
        }
    }
    __syncthreads();  // the coefficient rows are visible

    // ---- phase 2: the planes again (from cache), dx, the only write
#pragma unroll 1
    for (int h = 0; h < kWideRows / HALF; ++h) {
        if (h * HALF >= R) break;
        load_half(h);
#pragma unroll
        for (int rr = 0; rr < HALF; ++rr) {
            const int r = h * HALF + rr, n = wave * R + r;
            const bool ok = r < R && n < N;
            if (!ok) continue;
            const int ia = n * CH + wl.a, ib = n * CH + (wl.a + 1 < CH ? wl.a + 1 : wl.a);
            mono_forget(dg_[rr]);
            mono_forget(dx_[rr]);
            mask_row(rr, ia, ib);
            const float cG = ps1[ia], cX = ps2[ia], xr = pxr[ia], c0_ = pc0[ia];
            const float cG2 = ps1[ib], cX2 = ps2[ib], xr2 = pxr[ib], c02 = pc0[ib];
            float ov[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const bool first = q < wl.qs;
                ov[q] = fmaf(first ? cG : cG2, melem<T, VEC>(dg_[rr], q),
                             fmaf(first ? cX : cX2, melem<T, VEC>(dx_[rr], q) - (first ? xr : xr2), first ? c0_ : c02));
            }
            const size_t off = ((size_t)n * C + c0) * M;
            mstore<T, VEC>(__builtin_amdgcn_make_buffer_rsrc((void*)(dx + off), 0, gbytes, 0x00020000), voff, mpack<T, VEC>(ov));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace cnsn
