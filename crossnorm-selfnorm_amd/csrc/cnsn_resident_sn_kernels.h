// Cluster-resident kernels for SelfNorm ALONE (round 3): what every site of the ResNet-50 / WideResNet configurations
// runs on every step whose CrossNorm is idle (models/cnsn.py:130-150 behind :159-164; with the residual block's add and
// ReLU folded in: models/imagenet/resnet_cnsn.py:117-122).
//
// Without CrossNorm nothing pairs two planes of a channel: the only coupling is BatchNorm1d's batch mean / variance of
// z[n] = w0*mean[n] + w1*std[n] over the N instances (cnsn.py:121,138) — TWO numbers per channel — and, backward, the two
// batch sums of the BatchNorm backward.  The general kernels (cnsn_resident_kernels.h) hand every plane's statistics to
// every member of the cluster, which then repeats the algebra of all N planes: 2N floats gathered, N staged `saved` rows
// (27 KB of LDS at N = 256 in the backward), three workgroup barriers and a serial section of 2.5-3 us per item.  Here a
// member publishes the PARTIAL batch moments of its own planes (4 floats), gathers K = N / planes-per-workgroup partials
// (1-2 KB at N = 256), merges them (Chan) per wave, and finishes its own planes' algebra alone: one barrier, no staged rows,
// no LDS beyond the parked item.  That buys (a) a short chain from one publish to the next, (b) LDS for parking a whole item
// at three workgroups per CU, which is what lets the epilogue variants (a third tensor in flight) be pipelined at all.
//
// Schedule = the pipelined kernels' (cnsn_resident_pipe_kernels.h): item t parked (LDS + a few registers), item t+1 in
// flight into registers; gather t -> algebra t -> statistics of t+1, publish -> slot by slot: apply t and store, park
// t+1's slot, issue t+2's loads.
//
// Numerics: plane statistics exactly as the other resident kernels (two-pass from registers, float); z per plane in
// double from those; a member's partial = (mean, M2) of its planes' z in double, published as mean_hi + mean_lo + M2
// (floats); the merge and everything cancellation-prone in double.  The `saved` contract is the common one
// (cnsn_layout.h): either direction pairs with any other strategy.
#pragma once
#include "cnsn_resident_pipe_kernels.h"

namespace cnsn {

// wave-wide sum of doubles: DPP inside the 16-lane rows (both halves moved by v_mov_b32_dpp), v_readlane across the rows —
// a tenth of the latency of the ds_bpermute butterfly __shfl_xor compiles to (this sits on the publish-to-publish chain)
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);  // row_half_mirror
    v += dpp_d<0x140>(v);  // row_mirror
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

constexpr int kSnxVals = 4;  // floats a member publishes per exchange round

// planes of member k that exist
__device__ __forceinline__ int snx_count(int N, int own, int k) {
    const int left = N - k * own;
    return left < 0 ? 0 : (left < own ? left : own);
}

// per-plane state a wave keeps for its own planes between the phases of the pipeline.  It lives in LDS (one record per
// plane): these are wave-uniform scalars, and VGPRs are what the planes in flight need.  EVERY lane of the owning wave
// writes the (identical) record and reads it back, so each thread only ever reads what it wrote itself: a record
// written by lane 0 alone and read by its 63 neighbours is a data race in the compiler's memory model — lanes are
// independent threads there — and hipcc did forward lane 0's store past the branch and hoist the others' loads ABOVE
// it (wrong batch sums in the first version of these kernels).  Other waves read a record only behind a barrier.
struct SnxFwdState {
    double z;      // w0*mean + w1*std
    float mu, sg;  // plane mean, sqrt(var + eps_sn)
};
struct SnxBwdState {
    double mu_c, zh, dt;   // saved mean and normalised pre-activation; dL/d(pre-sigmoid)
    float g, sig_p, s1, s2;  // saved gate and std; sum G', sum G'*(X - float(mu_c))
};

__host__ __device__ inline size_t snx_fwd_lds_bytes(int K, int own, int parked_slots, int vec_bytes) {
    return (size_t)4 * 64 * parked_slots * vec_bytes  // parked item: [wave][slot][lane]
           + align16((size_t)K * kSnxVals * 4)        // vals[K][4]
           + (size_t)2 * own * sizeof(SnxFwdState)    // own planes of the parked item / the item in flight
           + align16((size_t)own * 4)                 // gates of the parked item's planes
           + 16;                                      // "this workgroup gave up" flag
}
__host__ __device__ inline size_t snx_bwd_lds_bytes(int K, int own, int parked_slots, int vec_bytes) {
    return (size_t)4 * 64 * parked_slots * vec_bytes + align16((size_t)K * kSnxVals * 4)
           + (size_t)3 * own * sizeof(SnxBwdState)    // items t, t+1, t+2
           + align16((size_t)own * 4 * 4)             // dx coefficients of the parked item's planes
           + 16;
}

// Chan merge of the K members' partials of channel c: every wave does it for itself (wave-uniform result)
//   vals[4l] + vals[4l+1] = mean of member l's planes, vals[4l+2] = their M2
__device__ __forceinline__ void snx_merge(const float* vals, int K, int N, int own, double inv_n, double& mean, double& var) {
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int l = lane; l < K; l += 64) s += (double)snx_count(N, own, l) * ((double)vals[4 * l] + (double)vals[4 * l + 1]);
    mean = wave_sum_d(s) * inv_n;
    double q = 0.0;
    for (int l = lane; l < K; l += 64) {
        const double dm = (double)vals[4 * l] + (double)vals[4 * l + 1] - mean;
        q += (double)vals[4 * l + 2] + (double)snx_count(N, own, l) * dm * dm;
    }
    var = wave_sum_d(q) * inv_n;  // biased, as BatchNorm normalises (cnsn.py:121)
    var = var > 0.0 ? var : 0.0;
}

// publish four floats of member (c, k): lanes 0..3 of the calling wave
__device__ __forceinline__ void snx_publish(unsigned long long* gran, size_t member, unsigned epoch, float v0, float v1,
                                            float v2, float v3) {
    const int lane = threadIdx.x & 63;
    if (epoch) {
        if (lane < 4) {
            const float v = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
            put_tagged(gran + member * 4 + lane, v, epoch);
        }
    } else if (lane < 2) {
        put_granule(gran + member * 2 + lane, lane == 0 ? v0 : v2, lane == 0 ? v1 : v3);
    }
}

// Workgroups per CU the kernels are compiled for, from the register slots (4 VGPRs each) of the item in flight:
// forward x [+ addend], backward G, x [+ addend].
constexpr int snx_fwd_inflight(int slots, bool epi) { return (epi ? 2 : 1) * slots; }
#ifndef SNX_W_SMALL
#define SNX_W_SMALL 4       // workgroups per CU asked for the smallest items (tuning builds: up to 6)
#endif
#ifndef SNX_FWD_W4_MAX
#define SNX_FWD_W4_MAX 8    // forward: items of at most this many slots in flight are compiled for 4 workgroups per CU
#endif
constexpr int snx_fwd_waves(int slots, bool epi, int elem_bytes = 4) {  // (16-bit: unpacking a vector costs 8 more registers)
    return snx_fwd_inflight(slots, epi) <= (elem_bytes == 2 ? 7 : 8) ? SNX_W_SMALL
           : snx_fwd_inflight(slots, epi) <= SNX_FWD_W4_MAX         ? 4
           : snx_fwd_inflight(slots, epi) <= 26                     ? 3
                                                                     : 2;
}
constexpr int snx_bwd_inflight(int slots2, bool epi) { return epi ? slots2 + slots2 / 2 : slots2; }  // slots2 = G and x slots
constexpr int snx_bwd_waves(int slots2, bool epi) { return snx_bwd_inflight(slots2, epi) <= 24 ? 3 : 2; }
// slots of the parked item that may stay in registers (the rest always goes to LDS)
constexpr int snx_fwd_keep(int slots) { return slots < kPipeKeep ? slots : kPipeKeep; }
constexpr int snx_bwd_keep(int slots2, bool epi) {
    return snx_bwd_waves(slots2, epi) == 3 ? (slots2 < 8 ? slots2 : (snx_bwd_inflight(slots2, epi) > 16 ? 4 : 8))
                                           : (slots2 < 13 ? slots2 : (snx_bwd_inflight(slots2, epi) > 32 ? 7 : 13));
}

// Workgroups that share a CU are not served alike: the hardware issues oldest-first, so the workgroup that arrived first
// on a CU (blockIdx < #CUs) runs its serial sections at full speed while the one that arrived last is starved — measured at
// (256,512,28,28) bf16: workgroups 0..63 finish their 11 items after 124 us, workgroups 640..703 their 10 items after 157
// us, and the chip idles towards the end (profiles/r03_sn_cluster.md).  Wave priority by arrival rank evens it out.
#ifndef SNX_PRIO_MODE
#define SNX_PRIO_MODE 0
#endif
__device__ __forceinline__ void snx_set_priority() {
#if SNX_PRIO_MODE == 1
    const int rank = (int)blockIdx.x / 256;  // (256 CUs: the r-th workgroup to arrive on its CU)
    if (rank == 1) __builtin_amdgcn_s_setprio(1);
    if (rank >= 2) __builtin_amdgcn_s_setprio(2);
#elif SNX_PRIO_MODE == 2
    const int rank = (int)blockIdx.x / 256;
    if (rank == 1) __builtin_amdgcn_s_setprio(2);
    if (rank >= 2) __builtin_amdgcn_s_setprio(3);
#elif SNX_PRIO_MODE == 3
    const int rank = (int)blockIdx.x / 256;
    if (rank >= 2) __builtin_amdgcn_s_setprio(1);
#endif
}

// ---- dynamic channel assignment ------------------------------------------------------------------------------------------
// A cluster is K fixed co-resident workgroups (blockIdx / K); WHICH channel it takes next is decided at run time.  Why:
// workgroups sharing a CU are not served alike — the one that arrived first runs its serial sections at full speed, the
// last one is starved (measured at (256,512,28,28) bf16: workgroups 0..63 finish 11 items after 124 us, workgroups
// 640..703 need 157 us for 10, and the chip idles towards the end; wave priorities by arrival rank change nothing:
// profiles/r03_sn_cluster.md).  With a static round-robin every cluster gets the same number of channels; here the first
// two rounds are static (cluster q takes channels q and q + #clusters); from then on the cluster's member 0 draws the
// channel of item s+2 from a counter when it publishes its partial of item s and sends it along in the partial's spare
// fourth float — the members learn it from the gather of item s they do anyway, exactly when the loads of item s+2 are
// due (a separate mailbox polled by every wave was tried first: the polling storm on one line per cluster slowed the
// kernels 3-40x).  Fast clusters simply take more.  Membership never changes: the residency argument of
// cnsn_resident_kernels.h holds as it is.
// MEASURED AND SWITCHED OFF (profiles/r03_sn_cluster.md): the draw is an atomic round trip on the publish-to-publish chain of
// a whole cluster.  56x56 classes (K = 32..64, few clusters): no change (+-1 %); 28x28 classes (K = 16, 48-64 clusters):
// 2-3x SLOWER even with the counter on a cache line of its own — fast clusters take more, but every cluster waits for its
// member 0's draw each item.  The static round-robin stays; the option is kept because the measurement is the
// documentation of where the end-of-kernel idling does NOT get fixed.
#ifndef SNX_DYNAMIC
#define SNX_DYNAMIC 0
#endif
constexpr int kNoChan = 0x00ffffff;  // "no more channels" (also what a timed-out mailbox read yields)

__device__ __forceinline__ unsigned long long sload_glc_u64(const void* p) {
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// one ticket of the launch's counter ({launch tag, count}: nothing to clear between launches).  It lives in the SECOND
// 128-byte line of the control block: the first one holds the time-out word every waiting wave polls — with the counter
// next to it a draw took 150-750 us (max 3 ms), the atomic starved by the readers of its line, and a whole cluster waits
// for its member 0 meanwhile (profiles/r03_sn_cluster.md).
__device__ __forceinline__ unsigned snx_ticket(unsigned* ctl, unsigned epoch) {
    gu64* ctr = (gu64*)(ctl + 32);
    unsigned long long v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const unsigned cnt = (unsigned)(v >> 32) == epoch ? (unsigned)v : 0u;
        const unsigned long long want = ((unsigned long long)epoch << 32) | (unsigned long long)(cnt + 1u);
        if (__hip_atomic_compare_exchange_strong(ctr, &v, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return cnt;
    }
}

// member 0 of a cluster (its wave 0): the channel of the cluster's item after next, out of the launch's counter
__device__ __forceinline__ int snx_draw_channel(unsigned* ctl, unsigned epoch, int nq, int C) {
    unsigned t = 0;
    if ((threadIdx.x & 63) == 0) t = snx_ticket(ctl, epoch);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    const long long ch = 2ll * nq + (long long)t;
    return ch < (long long)C ? (int)ch : kNoChan;
}

// ================================================================================================
// forward:  y = act(g[n,c] * (x [+ addend]))
// ================================================================================================
template <typename T, int VEC, int NV, int PPW, bool EPI>
__global__ __launch_bounds__(kBlock, snx_fwd_waves(PPW * NV, EPI, (int)sizeof(T))) void resident_sn_fwd_kernel(
    ResArgs ra, int npark, const T* __restrict__ x, const T* __restrict__ addend, int relu, T* __restrict__ y, GateDev gg,
    unsigned long long* __restrict__ gran, double* __restrict__ saved, unsigned* __restrict__ ctl) {
    constexpr int OWN = 4 * PPW;
    constexpr int SLOTS = PPW * NV;
    constexpr int KEEP = snx_fwd_keep(SLOTS), FIRST_KEEP = SLOTS - KEEP;
    constexpr int VB = VEC * (int)sizeof(T);
    const int NPARK = __builtin_amdgcn_readfirstlane(npark);  // FIRST_KEEP <= NPARK <= SLOTS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = ra.mid;
    const int N = a.N, C = a.C, K = ra.K;
    Raw<T, VEC>* park = (Raw<T, VEC>*)smem;
    float* vals = (float*)(smem + (size_t)4 * 64 * NPARK * VB);
    SnxFwdState* state = (SnxFwdState*)((char*)vals + align16((size_t)K * kSnxVals * 4));  // [2][OWN]
    float* gate = (float*)(state + 2 * OWN);
    int* gave_up = (int*)((char*)gate + align16((size_t)OWN * 4));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t P = (size_t)N * C;
    const SlotGeom<VEC, NV, false> sg(ra, lane);
    const int voff = lane * VB;
    Raw<T, VEC>* mypark = park + (size_t)wave * NPARK * 64 + lane;  // slot i of this lane: mypark[i * 64]
    // this workgroup is member k of cluster q_ (fixed); an "item" below is the channel the cluster works on
    const int q_ = (int)blockIdx.x / K, k = (int)blockIdx.x - q_ * K, nq = (int)gridDim.x / K;
    auto static_channel = [&](int seq) {
        const long long ch = (long long)q_ + (long long)seq * nq;
        return ch < (long long)C ? (int)ch : kNoChan;
    };

    if (threadIdx.x == 0) *gave_up = 0;
    __syncthreads();
    startup_skew(ra);
    snx_set_priority();

    Raw<T, VEC> d[PPW][NV];                       // the item in flight: x, then x + addend
    Raw<T, VEC> da[EPI ? PPW : 1][EPI ? NV : 1];  // its addend planes
    Raw<T, VEC> keep[KEEP];                       // slots of the parked item that did not go to LDS

    auto load_item = [&](int item) {
        const int c = __builtin_amdgcn_readfirstlane(item);  // (wave-uniform: plane bases and granule addresses stay scalar)
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const int n = (k * 4 + wave) * PPW + s;
            const size_t off = ((size_t)(n < N ? n : 0) * C + c) * ra.M;
            const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;  // past the batch end: every lane reads zeros
#pragma unroll
            for (int j = 0; j < NV; ++j) d[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(x + off, pbytes, j), voff);
            if constexpr (EPI) {
                const int abytes = addend ? pbytes : 0;
                const T* ab = addend ? addend + off : x;
#pragma unroll
                for (int j = 0; j < NV; ++j) da[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(ab, abytes, j), voff);
            }
        }
    };

    // statistics of the planes in d (exact two-pass from registers), z of each -> state[buf]; the member's partial batch
    // moments -> the cluster
    // draw: this cluster will work on the item after `item` too, so its member 0 draws the channel of the one after THAT
    auto stats_publish = [&](int item, int buf, bool draw) {
        const int c = __builtin_amdgcn_readfirstlane(item);
        const double w0 = gg.w[2 * c], w1 = gg.w[2 * c + 1];
        SnxFwdState* st = state + buf * OWN;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            if constexpr (EPI) {
                if (addend) {  // the op's input is x + addend, rounded to T like the reference's `out += identity`
#pragma unroll
                    for (int j = 0; j < NV; ++j) d[s][j] = add_raw<T, VEC>(d[s][j], da[s][j]);
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int q = 0; q < VEC; ++q) sum += elem<T, VEC>(d[s][j], q);
            const float mean = wave_sum(sum) / (float)ra.M;
            float m2 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (sg.valid(j)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float t = elem<T, VEC>(d[s][j], q) - mean;
                        m2 = fmaf(t, t, m2);
                    }
                }
            MomentsT<float> o;
            o.mu_c = o.mu_s = mean;
            o.M2c = o.M2s = wave_sum(m2);
            o.mu_o = o.M2o = 0.f;
            const FwdPlaneT<float> f = fwd_plane<float>(a, o, 0.f, 0.f);
            SnxFwdState r;  // (every lane: see SnxFwdState)
            r.z = w0 * (double)f.mu_p + w1 * (double)f.sig_p;
            r.mu = f.mu_p;
            r.sg = f.sig_p;
            st[wave * PPW + s] = r;
        }
        __syncthreads();
        if (wave == 0) {  // (wave-uniform arithmetic: every lane the same numbers)
            const int cnt = snx_count(N, OWN, k);
            double m = 0.0;
            for (int i = 0; i < cnt; ++i) m += st[i].z;
            m *= (double)__builtin_amdgcn_rcpf((float)cnt);  // (any point near the mean serves: M2 is taken about IT)
            const float m_hi = (float)m, m_lo = (float)(m - (double)m_hi);
            const double mp = (double)m_hi + (double)m_lo;
            double q = 0.0, r = 0.0;
            for (int i = 0; i < cnt; ++i) {
                const double t = st[i].z - mp;
                q += t * t;
                r += t;
            }
            // true mean of the member = mp + r/cnt, M2 about it = q - r*r/cnt (r is rounding-sized): fold into lo / M2
            const double corr = r * (double)__builtin_amdgcn_rcpf((float)cnt);
            float spare = __uint_as_float((unsigned)kNoChan);
#if SNX_DYNAMIC
            if (k == 0 && draw) spare = __uint_as_float((unsigned)snx_draw_channel(ctl, ra.epoch, nq, C));
#endif
            if (!(ra.fault && c == 0 && k == K - 1))
                snx_publish(gran, (size_t)c * K + k, ra.epoch, m_hi, (float)((double)m_lo + corr), (float)(q - r * corr), spare);
        }
    };
    auto park_item = [&]() {
#pragma unroll
        for (int s = 0; s < PPW; ++s)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = s * NV + j;
                if (i < FIRST_KEEP || i < NPARK)  // (wave-uniform)
                    mypark[i * 64] = d[s][j];
                else
                    keep[i - FIRST_KEEP] = d[s][j];
            }
    };

    int item = static_channel(0);
    if (item == kNoChan) return;   // (the grid never exceeds the items)
    int buf = 0;                   // state[buf]: the parked item; state[buf ^ 1]: the item in flight
    int iter_ = 0;                 // the cluster's item number
    int next = static_channel(1);
    load_item(item);
    stats_publish(item, buf, next != kNoChan);
    park_item();
    if (next != kNoChan) load_item(next);

    for (;;) {
        const int c = __builtin_amdgcn_readfirstlane(item);
        const bool more = next != kNoChan;  // workgroup-uniform
        CNSN_STAMP(0);

        // ---- per-channel parameters of item t (scalar loads: in flight during the gather)
        const float pgam = gg.gamma[c], pbet = gg.beta[c], prm = gg.run_mean[c], prv = gg.run_var[c];

        // ---- gather the K partials of item t's channel
        unsigned passes_ = 0;
        {
            const bool got = ra.epoch ? sweep_tagged_scalar(gran + (size_t)c * K * 4, K * 4, vals, ctl, ra.host_flag,
                                                            ra.wait_ticks, wave, ra.epoch, passes_)
                                      : sweep_granules_scalar(gran + (size_t)c * K * 2, K * 4, vals, ctl, ra.host_flag,
                                                              ra.wait_ticks, wave, passes_);
            if (lane == 0 && !got) *gave_up = 1;
        }
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out: see sweep_granules
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = (k * 4 + wave) * PPW + s;
                if (n < N && lane == 0) poison_plane<T, VEC>(y + ((size_t)n * C + c) * ra.M);
            }
            return;
        }
        CNSN_STAMP(1);
        CNSN_NOTE(6, passes_);

        // ---- BatchNorm1d over the batch from the merged partials; gates of this wave's planes
        {
            double mg, vg;
            snx_merge(vals, K, N, OWN, a.inv_n, mg, vg);
            // rstd to float accuracy: a uniform scale on the normalised value, and the SAME number reaches the backward
            const double rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
            if (k == 0 && threadIdx.x == 0) {
                const double mom_ = a.momentum;
                gg.run_mean[c] = (float)((1.0 - mom_) * (double)prm + mom_ * mg);
                gg.run_var[c] = (float)((1.0 - mom_) * (double)prv + mom_ * vg * a.unbias_n);
                if (saved) {
                    saved[SV_ROWS * P + c] = rg;
                    saved[SV_ROWS * P + C + c] = 1.0;
                }
            }
            {
#pragma unroll
                for (int s = 0; s < PPW; ++s) {
                    const int n = (k * 4 + wave) * PPW + s;
                    const SnxFwdState r = state[buf * OWN + wave * PPW + s];
                    const double zhg = (r.z - mg) * rg;
                    const float g = sigmoid_r<float>((float)((double)pgam * zhg + (double)pbet));
                    gate[wave * PPW + s] = g;  // (every lane)
                    if (saved && n < N && lane == 0) {
                        const SvRec p = sv_rec(n, c, N);
                        saved[sv_at(p, SV_MU_C)] = r.mu;
                        saved[sv_at(p, SV_MU_P)] = r.mu;
                        saved[sv_at(p, SV_SIG_P)] = r.sg;
                        saved[sv_at(p, SV_G)] = g;
                        saved[sv_at(p, SV_ZH_G)] = zhg;
                        saved[sv_at(p, SV_F)] = 1.0;
                        saved[sv_at(p, SV_ZH_F)] = 0.0;
                        if (a.save_coefs) store_fwd_coefs(saved, p, FwdCoefs{g, 0.f, 0.f, g, 0.f});
                    }
                }
            }
        }

        // ---- item t+1 has arrived long ago: its partial goes out BEFORE item t is applied (the barrier inside also
        //      separates this iteration's readers of vals from the next gather)
        // the channel of the cluster's item after next: it came with member 0's partial (static rounds: computed)
        int next2 = kNoChan;
        if (more) {
#if SNX_DYNAMIC
            next2 = (int)__float_as_uint(vals[3]);
#else
            next2 = static_channel(iter_ + 2);
#endif
        }
        next2 = __builtin_amdgcn_readfirstlane(next2);
        const bool more2 = next2 != kNoChan;
        CNSN_STAMP(2);
        if (more) stats_publish(next, buf ^ 1, more2);
        CNSN_STAMP(3);

        // ---- slot by slot: apply item t (the only write of y), park item t+1's slot in its place, send the loads of
        //      item t+2's slot after it.  y = fma(g, X - 0, 0): the one rounding of the reference's x * g (fwd_coefs)
        {
            const int c2 = next2, k2 = k;
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = (k * 4 + wave) * PPW + s;
                const float a_in = gate[wave * PPW + s];
                T* yb = y + ((size_t)(n < N ? n : 0) * C + c) * ra.M;
                const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;  // (a plane past the batch end drops its stores)
                const int n2 = (k2 * 4 + wave) * PPW + s;
                const bool live2 = more2 && n2 < N;
                const size_t off2 = ((size_t)(live2 ? n2 : 0) * C + (more2 ? c2 : 0)) * ra.M;
                const int pbytes2 = live2 ? ra.M * (int)sizeof(T) : 0;  // nothing to load: zeros, no traffic
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = s * NV + j;
                    Raw<T, VEC> v;
                    if (i < FIRST_KEEP || i < NPARK)
                        v = mypark[i * 64];
                    else
                        v = keep[i - FIRST_KEEP];
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        ov[q] = fmaf(a_in, elem<T, VEC>(v, q) - 0.f, 0.f);
                        if constexpr (EPI) ov[q] = relu ? fmaxf(ov[q], 0.f) : ov[q];
                    }
                    buf_store<T, VEC>(slot_rsrc<T, VEC>(yb, pbytes, j), voff, pack<T, VEC>(ov));
                    if (i < FIRST_KEEP || i < NPARK)  // park slot i of item t+1 (garbage after the last item: never read)
                        mypark[i * 64] = d[s][j];
                    else
                        keep[i - FIRST_KEEP] = d[s][j];
                    d[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(x + off2, pbytes2, j), voff);
                    if constexpr (EPI) {
                        const int abytes2 = addend ? pbytes2 : 0;
                        da[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(addend ? addend + off2 : x, abytes2, j), voff);
                    }
                }
            }
        }
        CNSN_STAMP(4);
        if (!more) break;
        buf ^= 1;
        item = next;
        next = next2;
        ++iter_;
    }
}

// ================================================================================================
// backward:  dx = g*G' + dmu/M + k*(X - mu),  G' = G masked by the forward's ReLU, X = x [+ addend]
// ================================================================================================
// Exchange: round A, the member's partial sums of dt and dt*zh (hi + lo floats each) — everybody gathers them; round B,
// every WAVE's partial sums of dz*mean and dz*std over its planes (the Conv1d taps' gradient, cnsn.py:119,137) — only ONE
// member per channel (rotating) gathers them, AFTER its stores are on their way: nobody else ever waits for them.
// Slot order of an item: plane s: G slots (s*2*NV + j), then X slots (s*2*NV + NV + j).
template <typename T, int VEC, int NV, int PPW, bool EPI>
__global__ __launch_bounds__(kBlock, snx_bwd_waves(2 * PPW * NV, EPI)) void resident_sn_bwd_kernel(
    ResArgs ra, int npark, const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ addend, int relu,
    T* __restrict__ dx, GateDev gg, GateGradDev dgr, unsigned long long* __restrict__ gran,
    unsigned long long* __restrict__ gran_b, const double* __restrict__ saved, unsigned* __restrict__ ctl) {
    constexpr int OWN = 4 * PPW;
    constexpr int SLOTS = 2 * PPW * NV;
    constexpr int KEEP = snx_bwd_keep(SLOTS, EPI), FIRST_KEEP = SLOTS - KEEP;
    constexpr int VB = VEC * (int)sizeof(T);
    const int NPARK = __builtin_amdgcn_readfirstlane(npark);  // FIRST_KEEP <= NPARK <= SLOTS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = ra.mid;
    const int N = a.N, C = a.C, K = ra.K;
    Raw<T, VEC>* park = (Raw<T, VEC>*)smem;
    float* vals = (float*)(smem + (size_t)4 * 64 * NPARK * VB);
    SnxBwdState* state = (SnxBwdState*)((char*)vals + align16((size_t)K * kSnxVals * 4));  // [3][OWN]
    float* coef = (float*)(state + 3 * OWN);                                                // [OWN][4]: cG, cX, xr, c0
    int* gave_up = (int*)((char*)coef + align16((size_t)OWN * 4 * 4));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t P = (size_t)N * C;
    const SlotGeom<VEC, NV, false> sg(ra, lane);
    const int voff = lane * VB;
    Raw<T, VEC>* mypark = park + (size_t)wave * NPARK * 64 + lane;

    if (threadIdx.x == 0) *gave_up = 0;
    __syncthreads();
    startup_skew(ra);
    snx_set_priority();

    Raw<T, VEC> dg_[PPW][NV], dx_[PPW][NV];       // the item in flight
    Raw<T, VEC> da[EPI ? PPW : 1][EPI ? NV : 1];  // its addend planes
    Raw<T, VEC> keep[KEEP > 0 ? KEEP : 1];        // slots of the parked item that did not go to LDS

    // `saved` rows of this wave's plane s of an item (wave-uniform loads) -> state[buf]
    auto fetch_rows = [&](int c, int n, int buf, int s) {
        const SvRec p = sv_rec((n < N ? n : 0), c, N);
        SnxBwdState r;
        r.mu_c = saved[sv_at(p, SV_MU_C)];
        r.zh = saved[sv_at(p, SV_ZH_G)];
        r.g = (float)saved[sv_at(p, SV_G)];
        r.sig_p = (float)saved[sv_at(p, SV_SIG_P)];
        r.dt = 0.0;
        r.s1 = r.s2 = 0.f;
        state[buf * OWN + wave * PPW + s] = r;  // (every lane: see SnxFwdState)
    };
    auto load_item = [&](int item, int buf) {
        const int c = item / K, k = item - c * K;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const int n = (k * 4 + wave) * PPW + s;
            fetch_rows(c, n, buf, s);
            const size_t off = ((size_t)(n < N ? n : 0) * C + c) * ra.M;
            const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) dg_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(gy + off, pbytes, j), voff);
#pragma unroll
            for (int j = 0; j < NV; ++j) dx_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(x + off, pbytes, j), voff);
            if constexpr (EPI) {
                const int abytes = addend ? pbytes : 0;
                const T* ab = addend ? addend + off : x;
#pragma unroll
                for (int j = 0; j < NV; ++j) da[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(ab, abytes, j), voff);
            }
        }
    };

    // per-plane sums of G' against X (shifted by the saved mean, as pass A' does), dt of each plane -> state[buf]; the
    // member's partial batch sums (round A) -> the cluster
    auto sums_publish = [&](int item, int buf) {
        const int c = item / K, k = item - c * K;
        SnxBwdState* st = state + buf * OWN;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const double mu_c = st[wave * PPW + s].mu_c;
            const float g = st[wave * PPW + s].g;
            if constexpr (EPI) {
                if (addend) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) dx_[s][j] = add_raw<T, VEC>(dx_[s][j], da[s][j]);
                }
                if (relu) {  // shut the gradient where the forward's output was not positive: the forward affine of a
                             // SelfNorm-only call is y = fma(float(g), X - 0, 0) whichever strategy ran it (fwd_coefs)
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        float gm[VEC];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const float t = fmaf(g, elem<T, VEC>(dx_[s][j], q) - 0.f, 0.f);
                            gm[q] = relu_open_r<T>(t) ? elem<T, VEC>(dg_[s][j], q) : 0.f;
                        }
                        dg_[s][j] = pack<T, VEC>(gm);
                    }
                }
            }
            const float si = (float)mu_c;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (sg.valid(j)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = elem<T, VEC>(dg_[s][j], q), X = elem<T, VEC>(dx_[s][j], q);
                        acc0 += G;
                        acc1 = fmaf(G, X - si, acc1);
                    }
                }
            const float s1 = wave_sum(acc0), s2 = wave_sum(acc1);
            const BwdSumsT<float> sm = fix_sums<float>(a, s1, s2, 0.f, 0.f, mu_c, 0.0);
            float dtg, dtf;
            gate_dt<float>(a, sm, 1.f, si, 0.f, si, g, 1.f, dtg, dtf);
            st[wave * PPW + s].s1 = s1;  // (every lane)
            st[wave * PPW + s].s2 = s2;
            st[wave * PPW + s].dt = (double)dtg;
        }
        __syncthreads();
        if (wave == 0) {  // (wave-uniform arithmetic)
            const int cnt = snx_count(N, OWN, k);
            double sa = 0.0, sb = 0.0;
            for (int i = 0; i < cnt; ++i) {
                sa += st[i].dt;
                sb += st[i].dt * st[i].zh;
            }
            const float a_hi = (float)sa, b_hi = (float)sb;
            if (!(ra.fault && item == K - 1))
                snx_publish(gran, (size_t)c * K + k, ra.epoch, a_hi, (float)(sa - (double)a_hi), b_hi,
                            (float)(sb - (double)b_hi));
        }
    };
    // registers -> LDS (+ keep): every lane writes (and later reads back) its own slots only
    auto park_item = [&]() {
#pragma unroll
        for (int s = 0; s < PPW; ++s)
#pragma unroll
            for (int j = 0; j < 2 * NV; ++j) {
                const int i = s * 2 * NV + j;
                const Raw<T, VEC> v = j < NV ? dg_[s][j] : dx_[s][j - NV];
                if (i < FIRST_KEEP || i < NPARK)  // (wave-uniform)
                    mypark[i * 64] = v;
                else
                    keep[i - FIRST_KEEP] = v;
            }
    };
    auto parked = [&](int i) -> Raw<T, VEC> {
        if (i < FIRST_KEEP || i < NPARK) return mypark[i * 64];
        return keep[i - FIRST_KEEP];
    };

    int item = blockIdx.x;
    if (item >= ra.items) return;
    int b0 = 0, b1 = 1, b2 = 2;  // state buffers of items t, t+1, t+2
    int iter_ = 0;
    (void)iter_;
    load_item(item, b0);
    sums_publish(item, b0);
    park_item();
    if (item + (int)gridDim.x < ra.items) load_item(item + (int)gridDim.x, b1);

    for (;;) {
        const int c = item / K, k = item - c * K;
        const int next = item + (int)gridDim.x, next2 = next + (int)gridDim.x;
        const bool more = next < ra.items, more2 = next2 < ra.items;  // workgroup-uniform
        CNSN_STAMP(0);

        // ---- per-channel parameters of item t (scalar loads: in flight during the gather)
        const float pw0 = gg.w[2 * c], pw1 = gg.w[2 * c + 1], pgam = gg.gamma[c];
        const double prs = saved[SV_ROWS * P + c];

        // ---- gather round A of item t's channel
        unsigned passes_ = 0;
        {
            const bool got = ra.epoch ? sweep_tagged_scalar(gran + (size_t)c * K * 4, K * 4, vals, ctl, ra.host_flag,
                                                            ra.wait_ticks, wave, ra.epoch, passes_)
                                      : sweep_granules_scalar(gran + (size_t)c * K * 2, K * 4, vals, ctl, ra.host_flag,
                                                              ra.wait_ticks, wave, passes_);
            if (lane == 0 && !got) *gave_up = 1;
        }
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = (k * 4 + wave) * PPW + s;
                if (n < N && lane == 0) poison_plane<T, VEC>(dx + ((size_t)n * C + c) * ra.M);
            }
            return;
        }
        CNSN_STAMP(1);
        CNSN_NOTE(6, passes_);

        // ---- batch sums of the BatchNorm backward; dx coefficients of this wave's planes; round B
        const bool reporter = k == c % K;  // the member that writes the channel's parameter gradients
        {
            BnBwd b{};
            double sa = 0.0, sb = 0.0;
            for (int l = lane; l < K; l += 64) {
                sa += (double)vals[4 * l] + (double)vals[4 * l + 1];
                sb += (double)vals[4 * l + 2] + (double)vals[4 * l + 3];
            }
            b.s_dt_g = wave_sum_d(sa);
            b.s_dtz_g = wave_sum_d(sb);
            b.wg0 = pw0;
            b.wg1 = pw1;
            b.kg = (double)pgam * prs;
            float pdw0 = 0.f, pdw1 = 0.f;
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = (k * 4 + wave) * PPW + s;
                const SnxBwdState r = state[b0 * OWN + wave * PPW + s];
                const float mu = (float)r.mu_c;
                const BwdSumsT<float> sm = fix_sums<float>(a, r.s1, r.s2, 0.f, 0.f, r.mu_c, 0.0);
                const BwdPlaneT<float> o =
                    bwd_plane<float>(a, b, sm, r.dt, 0.0, r.zh, 0.0, r.g, 1.f, 1.f, 1.f, mu, mu, r.sig_p, 1.f, 0.f);
                const BwdCoefs cf = bwd_coefs<float>(a, o, 0.f, 0.f, r.g, 1.f, mu, mu, r.mu_c, 1.f, r.mu_c, 1.f);
                {
                    float* oc = coef + (wave * PPW + s) * 4;  // (every lane)
                    oc[0] = cf.cG_in;
                    oc[1] = cf.cX_in;
                    oc[2] = cf.xr_in;
                    oc[3] = cf.c0_in;
                }
                if (n < N) {
                    pdw0 = fmaf(o.dz_g, mu, pdw0);
                    pdw1 = fmaf(o.dz_g, r.sig_p, pdw1);
                }
            }
            // round B: this wave's share of the taps' gradient (lanes 0 / 1)
            const size_t wm = ((size_t)c * K + k) * 4 + wave;
            if (ra.epoch) {
                if (lane < 2) put_tagged(gran_b + wm * 2 + lane, lane == 0 ? pdw0 : pdw1, ra.epoch);
            } else if (lane == 0) {
                put_granule(gran_b + wm, pdw0, pdw1);
            }
            if (reporter && threadIdx.x == 0) {
                dgr.dgamma[c] = (float)b.s_dtz_g;
                dgr.dbeta[c] = (float)b.s_dt_g;
            }
        }

        // ---- item t+1 has arrived long ago: its partial sums go out BEFORE item t is applied
        CNSN_STAMP(2);
        if (more) sums_publish(next, b1);
        CNSN_STAMP(3);

        // ---- slot by slot: apply item t (the only write of dx), park item t+1's slots in its place, send the loads of
        //      item t+2's slots after it
        {
            const int c2 = next2 / K, k2 = next2 - c2 * K;
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = (k * 4 + wave) * PPW + s;
                const float* oc = coef + (wave * PPW + s) * 4;
                const float cG = oc[0], cX = oc[1], xr = oc[2], c0 = oc[3];
                T* db = dx + ((size_t)(n < N ? n : 0) * C + c) * ra.M;
                const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;  // (a plane past the batch end drops its stores)
                const int n2 = (k2 * 4 + wave) * PPW + s;
                const bool live2 = more2 && n2 < N;
                fetch_rows(more2 ? c2 : 0, live2 ? n2 : 0, b2, s);
                const size_t off2 = ((size_t)(live2 ? n2 : 0) * C + (more2 ? c2 : 0)) * ra.M;
                const int pbytes2 = live2 ? ra.M * (int)sizeof(T) : 0;  // nothing to load: zeros, no traffic
                const int base = s * 2 * NV;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int ig = base + j, ix = base + NV + j;
                    const Raw<T, VEC> rg = parked(ig), rx = parked(ix);
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q)
                        ov[q] = fmaf(cG, elem<T, VEC>(rg, q), fmaf(cX, elem<T, VEC>(rx, q) - xr, c0));
                    buf_store<T, VEC>(slot_rsrc<T, VEC>(db, pbytes, j), voff, pack<T, VEC>(ov));
                    if (ig < FIRST_KEEP || ig < NPARK)
                        mypark[ig * 64] = dg_[s][j];
                    else
                        keep[ig - FIRST_KEEP] = dg_[s][j];
                    if (ix < FIRST_KEEP || ix < NPARK)
                        mypark[ix * 64] = dx_[s][j];
                    else
                        keep[ix - FIRST_KEEP] = dx_[s][j];
                    dg_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(gy + off2, pbytes2, j), voff);
                    dx_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(x + off2, pbytes2, j), voff);
                    if constexpr (EPI) {
                        const int abytes2 = addend ? pbytes2 : 0;
                        da[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(addend ? addend + off2 : x, abytes2, j), voff);
                    }
                }
            }
        }

        CNSN_STAMP(4);
        // ---- the reporter's wave 0 collects round B of channel c (4K wave shares), now that its stores are out
        if (reporter && wave == 0) {
            const int total = 4 * K;
            double s0 = 0.0, s1 = 0.0;
            bool failed = false;
            long long t_start = 0;
            for (int base = 0; base < total && !failed; base += 64) {
                const int i = base + lane;
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    float v0 = 0.f, v1 = 0.f;
                    if (i < total) {
                        const size_t wm = (size_t)c * K * 4 + i;
                        if (ra.epoch) {
                            const unsigned long long q0 =
                                __hip_atomic_load((gu64*)(gran_b + wm * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long q1 =
                                __hip_atomic_load((gu64*)(gran_b + wm * 2 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (unsigned)(q0 >> 32) == ra.epoch && (unsigned)(q1 >> 32) == ra.epoch;
                            v0 = __uint_as_float((unsigned)q0);
                            v1 = __uint_as_float((unsigned)q1);
                        } else {
                            const unsigned long long q0 =
                                __hip_atomic_load((gu64*)(gran_b + wm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = q0 != kGranuleEmpty;
                            v0 = __uint_as_float((unsigned)q0);
                            v1 = __uint_as_float((unsigned)(q0 >> 32));
                        }
                    }
                    if (__all(ok)) {
                        s0 += (double)v0;
                        s1 += (double)v1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(CNSN_POLL_SLEEP);
                    if ((spins & 15u) == 15u) {
                        const long long now = (long long)wall_clock64();
                        if (t_start == 0) t_start = now;
                        const unsigned seen = __hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen != ra.ctl_idle || now - t_start > ra.wait_ticks) {
                            if (seen == ra.ctl_idle && lane == 0) {
                                const unsigned prev =
                                    __hip_atomic_exchange((gu32*)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (prev == ra.ctl_idle && ra.host_flag)
                                    __hip_atomic_fetch_add(ra.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            failed = true;
                            break;
                        }
                    }
                }
            }
            s0 = wave_sum_d(s0);
            s1 = wave_sum_d(s1);
            if (lane == 0) {
                dgr.dw[2 * c] = failed ? __builtin_nanf("") : (float)s0;
                dgr.dw[2 * c + 1] = failed ? __builtin_nanf("") : (float)s1;
                if (failed) *gave_up = 1;
            }
        }
        if (!more) break;
        {
            const int t = b0;
            b0 = b1;
            b1 = b2;
            b2 = t;
        }
        item = next;
        ++iter_;
    }
}

}  // namespace cnsn
