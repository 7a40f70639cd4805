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
// t+1's slot, issue t+2's loads.  An item is a channel; a cluster is K fixed workgroups; a wave holds PPW planes of it
// (1 at 56x56 ... 16 for one-slot planes such as 14x14 in 16 bits).
//
// Register discipline: one buffer descriptor per TENSOR (PlaneIo), kernel arguments re-read from the
// kernarg segment where they are used (kargs_now), plane algebra lane-parallel, phase fences for the register allocator
// around the plane loops (snx_phase_fence) — without them a third of the hot loops' vector instructions were SGPR spills.
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

// put_lane with the lane as a (constant after unrolling) function argument
__device__ __forceinline__ int put_lane_at(int sval, int lane, int old) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(lane));
    return old;
}

constexpr int kSnxVals = 4;  // floats a member publishes per exchange round

// planes of member k that exist
__device__ __forceinline__ int snx_count(int N, int own, int k) {
    const int left = N - k * own;
    return left < 0 ? 0 : (left < own ? left : own);
}

// per-plane state a wave keeps for its own planes between the phases of the pipeline.  It lives in LDS (one record per
// plane).  The record of plane s of a wave is written by LANE s of that wave, which does the plane's scalar algebra (all
// planes of the wave at once, lane-parallel), and is read back by the same lane in the next phase — or by wave 0 behind a
// workgroup barrier.  Never "lane 0 writes, its 63 neighbours read": that is a data race in the compiler's memory model
// — lanes are independent threads there — and hipcc did forward lane 0's store past the branch and hoist the others'
// loads ABOVE it (wrong batch sums in the first version of these kernels).  What the plane loops need from a record
// (gate, dx coefficients, saved mean) travels from lane s to the whole wave through v_readlane, not through LDS.
struct SnxFwdState {
    double z;      // w0*mean + w1*std
    float mu, sg;  // plane mean, sqrt(var + eps_sn)
};
struct SnxBwdState {
    double mu_c, zh, dt;   // saved mean and normalised pre-activation; dL/d(pre-sigmoid)
    float g, sig_p, s1, s2;  // saved gate and std; sum G', sum G'*(X - float(mu_c))
};
// the CrossNorm rows of `saved` one plane's backward algebra needs (un-boxed call: cnsn_layout.h, SV_MU_P .. SV_M_IN)
struct SnxCnRows {
    double mu_s;
    float mu_p, aa, a1, m_in, sig_c, M2c, sig_s;
    float mu_o, s1o, s2o;  // with crop boxes: the saved mean outside the content box; sum G, sum G*(x - float(mu_o)) outside it
};
// CN variant of the backward: the record of an own plane carries its CrossNorm rows next to the SelfNorm ones
// ... and the rows of the plane that BORROWED this plane's statistics (same channel, instance perm^-1[n])
struct SnxBorrower {
    double mu_c, zh;
    float g, sig_p, mu_p, aa, a1, m_in, sig_c, M2c, mu_o;
};
struct SnxBwdStateCn {
    SnxBwdState sn;
    SnxCnRows cn;
};
__device__ __forceinline__ const SnxBwdState& snx_sn(const SnxBwdState& r) { return r; }
__device__ __forceinline__ const SnxBwdState& snx_sn(const SnxBwdStateCn& r) { return r.sn; }

__host__ __device__ inline size_t snx_fwd_lds_bytes(int K, int own, int parked_slots, int vec_bytes) {
    return (size_t)4 * 64 * parked_slots * vec_bytes  // parked item: [wave][slot][lane]
           + align16((size_t)K * kSnxVals * 4)        // vals[K][4]
           + (size_t)2 * own * sizeof(SnxFwdState)    // own planes of the parked item / the item in flight
           + 16;                                      // "this workgroup gave up" flag
}
__host__ __device__ inline size_t snx_bwd_lds_bytes(int K, int own, int parked_slots, int vec_bytes, bool cn = false, int N = 0) {
    return (size_t)4 * 64 * parked_slots * vec_bytes + align16((size_t)K * kSnxVals * 4)
           + (size_t)2 * own * (cn ? sizeof(SnxBwdStateCn) : sizeof(SnxBwdState))  // items t, t+1 (the rows of item t+2 wait in registers)
           + 16 + (cn ? align16((size_t)N * 4) : 0);  // (CN: the inverse batch permutation)
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
// (slots of 8-byte vectors — one-slot planes such as 14x14 in 16 bits — count half: vb = bytes per vector)
constexpr int snx_fwd_inflight(int slots, bool epi, int vb = 16) { return (epi ? 2 : 1) * slots * vb / 16; }
#ifndef SNX_W_SMALL
#define SNX_W_SMALL 4       // workgroups per CU asked for the smallest items (tuning builds: up to 6)
#endif
#ifndef SNX_FWD_W4_MAX
#define SNX_FWD_W4_MAX 8    // forward: items of at most this many slots in flight are compiled for 4 workgroups per CU
#endif
constexpr int snx_fwd_waves(int slots, bool epi, int elem_bytes = 4, int vb = 16) {  // (16-bit: unpacking a vector costs 8 more registers)
    return snx_fwd_inflight(slots, epi, vb) <= (elem_bytes == 2 ? 7 : 8) ? SNX_W_SMALL
           : snx_fwd_inflight(slots, epi, vb) <= SNX_FWD_W4_MAX         ? 4
           : snx_fwd_inflight(slots, epi, vb) <= 26                     ? 3
                                                                         : 2;
}
constexpr int snx_bwd_inflight(int slots2, bool epi, int vb = 16) {  // slots2 = G and x slots
    return (epi ? slots2 + slots2 / 2 : slots2) * vb / 16;
}
constexpr int snx_bwd_waves(int slots2, bool epi, int vb = 16) { return snx_bwd_inflight(slots2, epi, vb) <= 24 ? 3 : 2; }
// slots of the parked item that may stay in registers (the rest always goes to LDS)
constexpr int snx_min(int a, int b) { return a < b ? a : b; }
// (epi16: the 16-bit block forward of the 13-slot class spilled 3 VGPRs with six kept slots)
constexpr int snx_fwd_keep(int slots, int vb = 16, bool epi16 = false) { return snx_min(slots, kPipeKeep * 16 / vb) - ((epi16 && slots == 13) ? 1 : 0); }
#ifndef SNX_CN_KEEP_LESS
#define SNX_CN_KEEP_LESS 1
#endif
#ifndef SNX_CNB_KEEP_LESS
#define SNX_CNB_KEEP_LESS 3
#endif
// cn / boxed: the CrossNorm-capable backward holds the lender / borrower records next to the item — that many slots fewer stay in
// registers (round 5: those instantiations spilled 3-25 VGPRs to scratch memory; the build now refuses scratch in this family)
constexpr int snx_bwd_keep(int slots2, bool epi, int vb = 16, bool cn = false, bool boxed = false) {
    const int base = snx_min(slots2, (snx_bwd_waves(slots2, epi, vb) == 3 ? (snx_bwd_inflight(slots2, epi, vb) > 16 ? 4 : 8)
                                                                       : (snx_bwd_inflight(slots2, epi, vb) > 32 ? 7 : 13)) *
                                     16 / vb);
    // (the 13-slot class never spilled and is the headline's: untouched; the 8-slot class needs two slots more; the 16-slot
    //  class has no LDS left for another parked slot at two workgroups per CU: it keeps its spills, __graft_entry__.py)
    //  — and neither has the 8-slot class WITH crop boxes at three workgroups per CU)
    const int less = !cn || slots2 == 26 || slots2 == 32 || (slots2 == 16 && boxed)
                         ? 0
                         : (boxed ? SNX_CNB_KEEP_LESS : SNX_CN_KEEP_LESS) + (slots2 == 16 ? 2 : 0);
    // (the 16-bit 2-slot x 4-plane block backward still spills 5 VGPRs: with fewer kept slots its parked item no longer fits
    //  three workgroups' LDS — the one known exception of the family's no-scratch rule, __graft_entry__.py)
    return base > less ? base - less : 0;
}

// Workgroups that share a CU are not served alike: the hardware issues oldest-first, so the workgroup that arrived first
// on a CU (blockIdx < #CUs) runs its serial sections at full speed while the one that arrived last is starved — measured at
// (256,512,28,28) bf16: workgroups 0..63 finish their 11 items after 124 us, workgroups 640..703 their 10 items after 157
// us, and the chip idles towards the end (profiles/r03_sn_cluster.md).  Wave priority by arrival rank evens it out.
#ifndef SNX_PRIO_MODE
#define SNX_PRIO_MODE 0
#endif
// the serial section of an item (gather, algebra, sums of t+1) at wave priority 3, ahead of the neighbours' apply loops — where it
// measured faster (profiles/r05_serial_priority.md): the fp32 forward (-2 %) and the backward with crop boxes (-6 %); the 16-bit
// forward got 5 % SLOWER with it
#ifndef SNX_SERIAL_PRIO
#define SNX_SERIAL_PRIO 1
#endif
template <typename T>
constexpr bool snx_fwd_serial_prio() { return SNX_SERIAL_PRIO && sizeof(T) == 4; }
constexpr bool snx_bwd_serial_prio(bool boxed) { return SNX_SERIAL_PRIO && boxed; }
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
template <typename T>
struct SnxFwdKargs {
    ResArgs ra;
    int npark;
    const T* x;
    const T* addend;  // PRE addend or null
    int relu;
    T* y;
    GateDev gg;
    unsigned long long* gran;
    double* saved;
    unsigned* ctl;
};

template <typename T, int VEC, int NV, int PPW, bool EPI>
__global__ __launch_bounds__(kBlock, snx_fwd_waves(PPW * NV, EPI, (int)sizeof(T), VEC * (int)sizeof(T))) void resident_sn_fwd_kernel(
    SnxFwdKargs<T>) {
    using KA = SnxFwdKargs<T>;
#define KA_ (kargs_now<KA>())
    constexpr int OWN = 4 * PPW;
    constexpr int SLOTS = PPW * NV;
    constexpr int VB = VEC * (int)sizeof(T);
    constexpr int KEEP = snx_fwd_keep(SLOTS, VB, EPI && sizeof(T) == 2), FIRST_KEEP = SLOTS - KEEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // resident for the whole kernel: the geometry, the tensor descriptors, the item bookkeeping
    const KA* ka0 = KA_;
    const int NPARK_ = __builtin_amdgcn_readfirstlane(ka0->npark), NPARK = NPARK_;  // FIRST_KEEP <= NPARK <= SLOTS
    const int N = ka0->ra.mid.N, C = ka0->ra.mid.C, K = ka0->ra.K, M = ka0->ra.M;
    const bool has_add = EPI && ka0->addend != nullptr;
    Raw<T, VEC>* park = (Raw<T, VEC>*)smem;
    float* vals = (float*)(smem + (size_t)4 * 64 * NPARK * VB);
    SnxFwdState* state = (SnxFwdState*)((char*)vals + align16((size_t)K * kSnxVals * 4));  // [2][OWN]
    int* gave_up = (int*)(state + 2 * OWN);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const PlaneIo<T, VEC, NV> sg(ka0->ra, N, C, lane);
    const __amdgpu_buffer_rsrc_t rx = sg.tensor(ka0->x), ry = sg.tensor(ka0->y), radd = sg.tensor(has_add ? ka0->addend : ka0->x);
    Raw<T, VEC>* mypark = park + (size_t)wave * NPARK * 64 + lane;  // slot i of this lane: mypark[i * 64]
    // this workgroup is member k of cluster q_ (fixed); an "item" below is the channel the cluster works on
    const int vb_ = cluster_block(ka0->ra.xcd);  // (XCD-aware numbering: the cluster's members share one L2)
    const int q_ = vb_ / K, k = vb_ - q_ * K, nq = (int)gridDim.x / K;
    auto static_channel = [&](int seq) {
        const long long ch = (long long)q_ + (long long)seq * nq;
        return ch < (long long)C ? (int)ch : kNoChan;
    };
    // this wave's planes of ANY item: n0 .. n0 + PPW - 1, the first `nlive` of them inside the batch
    const int n0 = (k * 4 + wave) * PPW;
    const int nlive = N - n0 < 0 ? 0 : (N - n0 > PPW ? PPW : N - n0);
    const unsigned stride = (unsigned)C * (unsigned)M * (unsigned)sizeof(T);
#ifdef CNSN_PROF
    struct {
        unsigned long long* prof;
    } ra{ka0->ra.prof};
#endif

    if (threadIdx.x == 0) *gave_up = 0;
    __syncthreads();
    startup_skew(ka0->ra);
    snx_set_priority();

    Raw<T, VEC> d[PPW][NV];                       // the item in flight: x, then x + addend
    Raw<T, VEC> da[EPI ? PPW : 1][EPI ? NV : 1];  // its addend planes
    Raw<T, VEC> keep[KEEP];                       // slots of the parked item that did not go to LDS

    auto load_item = [&](int item) {
        const int c = __builtin_amdgcn_readfirstlane(item);  // (wave-uniform: plane offsets and granule addresses stay scalar)
        const unsigned span = sg.span(n0, c, C, M, true);
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const unsigned off = sg.at(span, stride, s, nlive);  // past the batch end: every lane reads zeros
#pragma unroll
            for (int j = 0; j < NV; ++j) d[s][j] = sg.load(rx, off, j);
            if constexpr (EPI) {
                const unsigned aoff = has_add ? off : sg.dead;
#pragma unroll
                for (int j = 0; j < NV; ++j) da[s][j] = sg.load(radd, aoff, j);
            }
        }
    };

    // statistics of the planes in d (exact two-pass from registers), z of each -> state[buf]; the member's partial batch
    // moments -> the cluster
    // draw: this cluster will work on the item after `item` too, so its member 0 draws the channel of the one after THAT
    auto stats_publish = [&](int item, int buf, bool draw) {
        const int c = __builtin_amdgcn_readfirstlane(item);
        snx_phase_fence();
        SnxFwdState* st = state + buf * OWN;
        int my_sum_b = 0, my_m2_b = 0;  // lane s < PPW: the sums of this wave's plane s, as bits (v_writelane puts them there:
                                        // a select per plane would keep one lane mask per plane alive across the item loop)
        const int M = KA_->ra.M;
        const PlaneIo<T, VEC, NV> sg(KA_->ra, 1, 1, lane);  // (only the slot validity is used here)
        const bool has_add = EPI && KA_->addend != nullptr;
        const float inv_m = __builtin_amdgcn_rcpf((float)M);
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            if constexpr (EPI) {
                if (has_add) {  // the op's input is x + addend, rounded to T like the reference's `out += identity`
#pragma unroll
                    for (int j = 0; j < NV; ++j) d[s][j] = add_raw<T, VEC>(d[s][j], da[s][j]);
                }
            }
            float sum = 0.f;
            if constexpr (CNSN_DOT2 && sizeof(T) == 2) {  // 16 bits: the plane sum from the packed words (no unpacking)
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int w = 0; w < VEC / 2; ++w) sum = dot2_acc<T>((unsigned)d[s][j][w], ones2<T>(), sum);
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) sum += elem<T, VEC>(d[s][j], q);
            }
            const int tot_b = wave_sum_bits(sum);
            my_sum_b = put_lane_at(tot_b, s, my_sum_b);
            const float mean = __int_as_float(tot_b) * inv_m;  // the shift of the second pass (M2 about a point one ulp off the mean is the same number)
            float m2 = 0.f;
            if constexpr (CNSN_DOT2 && sizeof(T) == 2) {  // ... and the second pass on element PAIRS (v_pk_add_f32 / v_pk_fma_f32)
                cnsn_f2_t m2v = {0.f, 0.f};
                const cnsn_f2_t mean2 = {mean, mean};
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (sg.valid(j)) {
#pragma unroll
                        for (int w = 0; w < VEC / 2; ++w) {
                            const cnsn_f2_t t = unpack2<T>((unsigned)d[s][j][w]) - mean2;
                            m2v = __builtin_elementwise_fma(t, t, m2v);
                        }
                    }
                m2 = m2v.x + m2v.y;
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (sg.valid(j)) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const float t = elem<T, VEC>(d[s][j], q) - mean;
                            m2 = fmaf(t, t, m2);
                        }
                    }
            }
            my_m2_b = put_lane_at(wave_sum_bits(m2), s, my_m2_b);
        }
        {   // lane s < PPW: the plane algebra of plane s — one pass of arithmetic whatever PPW is.  The record is written by
            // lane s alone and read back by the same lane, or by wave 0 behind the barrier (the lanes past PPW compute on
            // zeros and keep their results to themselves)
            const float my_sum = __int_as_float(my_sum_b), my_m2 = __int_as_float(my_m2_b);
            const KA* ka = KA_;
            const MidArgs a = ka->ra.mid;
            const double w0 = ka->gg.w[2 * c], w1 = ka->gg.w[2 * c + 1];
            MomentsT<float> o;
            o.mu_c = o.mu_s = my_sum / (float)M;
            o.M2c = o.M2s = my_m2;
            o.mu_o = o.M2o = 0.f;
            const FwdPlaneT<float> f = fwd_plane<float>(a, o, 0.f, 0.f);
            SnxFwdState r;
            r.z = w0 * (double)f.mu_p + w1 * (double)f.sig_p;
            r.mu = f.mu_p;
            r.sg = f.sig_p;
            if (lane < PPW) st[wave * PPW + lane] = r;
        }
        __syncthreads();
        if (wave == 0) {  // (wave-uniform arithmetic: every lane the same numbers)
            const int cnt = snx_count(N, OWN, k);
            double m = 0.0, q = 0.0, r = 0.0, mp;
            float m_hi, m_lo;
            if constexpr (OWN <= 8) {  // a handful of planes: a short serial loop beats three wave-wide sums
                for (int i = 0; i < cnt; ++i) m += st[i].z;
                m *= (double)__builtin_amdgcn_rcpf((float)cnt);  // (any point near the mean serves: M2 is taken about IT)
                m_hi = (float)m, m_lo = (float)(m - (double)m_hi);
                mp = (double)m_hi + (double)m_lo;
                for (int i = 0; i < cnt; ++i) {
                    const double t = st[i].z - mp;
                    q += t * t;
                    r += t;
                }
            } else {  // lane i takes plane i of the member (OWN <= 64)
                const double zi = lane < cnt ? st[lane < OWN ? lane : 0].z : 0.0;
                m = wave_sum_d(zi) * (double)__builtin_amdgcn_rcpf((float)cnt);
                m_hi = (float)m, m_lo = (float)(m - (double)m_hi);
                mp = (double)m_hi + (double)m_lo;
                const double t = lane < cnt ? zi - mp : 0.0;
                q = wave_sum_d(t * t);
                r = wave_sum_d(t);
            }
            // true mean of the member = mp + r/cnt, M2 about it = q - r*r/cnt (r is rounding-sized): fold into lo / M2
            const double corr = r * (double)__builtin_amdgcn_rcpf((float)cnt);
            float spare = __uint_as_float((unsigned)kNoChan);
            const KA* ka = KA_;
#if SNX_DYNAMIC
            if (k == 0 && draw) spare = __uint_as_float((unsigned)snx_draw_channel(ka->ctl, ka->ra.epoch, nq, C));
#else
            (void)draw;
#endif
            if (!(ka->ra.fault && c == 0 && k == K - 1))
                snx_publish(ka->gran, (size_t)c * K + k, ka->ra.epoch, m_hi, (float)((double)m_lo + corr), (float)(q - r * corr),
                            spare);
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
        if constexpr (snx_fwd_serial_prio<T>()) __builtin_amdgcn_s_setprio(3);  // the serial section ahead of the neighbours' apply loops
        const int c = __builtin_amdgcn_readfirstlane(item);
        const bool more = next != kNoChan;  // workgroup-uniform
        CNSN_STAMP(0);

        // ---- per-channel parameters of item t (scalar loads: in flight during the gather)
        float pgam, pbet, prm, prv;
        {
            const GateDev gg = KA_->gg;
            pgam = gg.gamma[c], pbet = gg.beta[c], prm = gg.run_mean[c], prv = gg.run_var[c];
        }
        float gate_l = 0.f;  // lane s: the gate of this wave's plane s of item t

        // ---- gather the K partials of item t's channel
        unsigned passes_ = 0;
        {
            const KA* ka = KA_;
            const unsigned epoch = ka->ra.epoch;
            const bool got = epoch ? sweep_tagged_scalar(ka->gran + (size_t)c * K * 4, K * 4, vals, ka->ctl, ka->ra.host_flag,
                                                         ka->ra.wait_ticks, wave, epoch, passes_)
                                   : sweep_granules_scalar(ka->gran + (size_t)c * K * 2, K * 4, vals, ka->ctl, ka->ra.host_flag,
                                                           ka->ra.wait_ticks, wave, passes_);
            if (lane == 0 && !got) *gave_up = 1;
        }
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out: see sweep_granules
            T* y = KA_->y;
            const int first = opaque_s(n0);  // (a cold path: nothing of it is worth computing ahead of the loop)
            for (int s = 0; s < PPW; ++s) {
                const int n = first + s;
                if (n < N && lane == 0) poison_plane<T, VEC>(y + ((size_t)n * C + c) * M);
            }
            return;
        }
        CNSN_STAMP(1);
        CNSN_NOTE(6, passes_);

        // ---- BatchNorm1d over the batch from the merged partials; gates of this wave's planes
        {
            const KA* ka = KA_;
            const MidArgs a = ka->ra.mid;
            double* saved = ka->saved;
            double mg, vg;
            snx_merge(vals, K, N, OWN, a.inv_n, mg, vg);
            // rstd to float accuracy: a uniform scale on the normalised value, and the SAME number reaches the backward
            const double rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
            if (k == 0 && threadIdx.x == 0) {
                const double mom_ = a.momentum;
                ka->gg.run_mean[c] = (float)((1.0 - mom_) * (double)prm + mom_ * mg);
                ka->gg.run_var[c] = (float)((1.0 - mom_) * (double)prv + mom_ * vg * a.unbias_n);
                if (c == 0) bump_batches_tracked(ka->gg.nbt);
                if (saved) {
                    const size_t P = (size_t)N * C;
                    saved[SV_ROWS * P + c] = rg;
                    saved[SV_ROWS * P + C + c] = 1.0;
                }
            }
            {   // lane s < PPW takes plane s of this wave (the others repeat plane 0): the gate stays in a register and is
                // handed to the apply loop with v_readlane — no LDS, nothing another lane wrote is read
                const int sl = lane < PPW ? lane : 0;
                const int n = (k * 4 + wave) * PPW + sl;
                const SnxFwdState r = state[buf * OWN + wave * PPW + sl];
                const double zhg = (r.z - mg) * rg;
                gate_l = sigmoid_r<float>((float)((double)pgam * zhg + (double)pbet));
                if (saved && n < N && lane < PPW) {   // (consecutive lanes: consecutive doubles of every row)
                    const SvRec p = sv_rec(n, c, N);
                    saved[sv_at(p, SV_MU_C)] = r.mu;
                    saved[sv_at(p, SV_MU_P)] = r.mu;
                    saved[sv_at(p, SV_SIG_P)] = r.sg;
                    saved[sv_at(p, SV_G)] = gate_l;
                    saved[sv_at(p, SV_ZH_G)] = zhg;
                    saved[sv_at(p, SV_F)] = 1.0;
                    saved[sv_at(p, SV_ZH_F)] = 0.0;
                    if (a.save_coefs) store_fwd_coefs(saved, p, FwdCoefs{gate_l, 0.f, 0.f, gate_l, 0.f});
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
        if constexpr (snx_fwd_serial_prio<T>()) __builtin_amdgcn_s_setprio(0);
        CNSN_STAMP(3);

        // ---- slot by slot: apply item t (the only write of y), park item t+1's slot in its place, send the loads of
        //      item t+2's slot after it.  y = fma(g, X - 0, 0): the one rounding of the reference's x * g (fwd_coefs)
        {
            // (a phase of its own for the register allocator: everything the plane loop touches is defined AFTER the fence,
            //  from the kernel arguments — a dozen scalar loads — so that it lives in registers for the length of the loop)
            snx_phase_fence();
            const KA* ka = KA_;
            const int relu = EPI ? ka->relu : 0;
            const int N = ka->ra.mid.N, C = ka->ra.mid.C, M = ka->ra.M;
            const PlaneIo<T, VEC, NV> sg(ka->ra, N, C, lane);
            const bool has_add = EPI && ka->addend != nullptr;
            const __amdgpu_buffer_rsrc_t rx = sg.tensor(ka->x), ry = sg.tensor(ka->y), radd = sg.tensor(has_add ? ka->addend : ka->x);
            const unsigned stride_ = (unsigned)C * (unsigned)M * (unsigned)sizeof(T);
            const int n0_ = opaque_s(n0);
            const int nlive_ = N - n0_ < 0 ? 0 : (N - n0_ > PPW ? PPW : N - n0_);
            const unsigned span = sg.span(n0_, c, C, M, true), span2 = sg.span(n0_, next2, C, M, more2);
            const int NPARK = __builtin_amdgcn_readfirstlane(ka->npark);  // (the parked-or-kept choices of the slots)
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const float a_in = lane_bcast(gate_l, s);
                const unsigned yoff = sg.at(span, stride_, s, nlive_);   // (a plane past the batch end drops its stores)
                const unsigned off2 = sg.at(span2, stride_, s, nlive_);  // nothing to load: zeros, no traffic
                const unsigned aoff2 = has_add ? off2 : sg.dead;
                (void)aoff2;
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
                    sg.store(ry, yoff, j, pack<T, VEC>(ov));
                    if (i < FIRST_KEEP || i < NPARK)  // park slot i of item t+1 (garbage after the last item: never read)
                        mypark[i * 64] = d[s][j];
                    else
                        keep[i - FIRST_KEEP] = d[s][j];
                    d[s][j] = sg.load(rx, off2, j);
                    if constexpr (EPI) da[s][j] = sg.load(radd, aoff2, j);
                }
            }
            snx_phase_fence();
        }
        CNSN_STAMP(4);
        if (!more) break;
        buf ^= 1;
        item = next;
        next = next2;
        ++iter_;
    }
#undef KA_
}

// ================================================================================================
// backward:  dx = g*G' + dmu/M + k*(X - mu),  G' = G masked by the forward's ReLU, X = x [+ addend]
// ================================================================================================
// Exchange: round A, the member's partial sums of dt and dt*zh (hi + lo floats each) — everybody gathers them; round B,
// every WAVE's partial sums of dz*mean and dz*std over its planes (the Conv1d taps' gradient, cnsn.py:119,137) — only ONE
// member per channel (rotating) gathers them, AFTER its stores are on their way: nobody else ever waits for them.
// Slot order of an item: plane s: G slots (s*2*NV + j), then X slots (s*2*NV + NV + j).
// The `saved` rows of an item's planes are fetched lane-parallel (lane s: plane s of this wave) together with the item's
// planes, two items ahead, and wait in registers until the item's sums are formed.
template <typename T>
struct SnxBwdKargs {
    ResArgs ra;
    int npark;
    const T* gy;
    const T* x;
    const T* addend;  // PRE addend or null
    int relu;
    T* dx;
    GateDev gg;
    GateGradDev dgr;
    unsigned long long* gran;
    unsigned long long* gran_b;
    const double* saved;
    unsigned* ctl;
};
// CN variant: + the per-plane sums every member publishes for the plane that LENT it its statistics, the batch permutation
// (device array, or as a launch argument: PermInline)
template <typename T>
struct SnxBwdKargsCn {
    SnxBwdKargs<T> sn;
    unsigned long long* gran_p;  // (C, N) planes x {sum G, sum G*(x - float(mu_c))}: 2 tagged granules or 1 untagged pair each
    const int64_t* perm;
    PermInline pin;
};

// ---- CrossNorm (un-boxed) in front of SelfNorm, backward (round 4) ------------------------------------------------------------
// With CrossNorm a plane (n, c) takes its statistics from plane q = perm[n] of the same channel (models/cnsn.py:62-68, the
// style source is not detached), so its gradient has a second contribution: what the plane r = perm^-1[n] that BORROWED
// (n, c)'s statistics sends back (Emu, Esig of bwd_plane).  The general cluster kernels therefore hand every plane's two sums
// to every member, stage the 15 `saved` rows of all N planes (30 KB per item and member: 0.5 GB of L2 reads per backward at
// the north-star shape) and repeat the algebra of the whole channel in every member.  None of that is needed: the only
// batch-wide quantities are still the two sums of the BatchNorm1d backward, which are sums of per-plane terms — a member
// publishes its PARTIAL of them exactly as without CrossNorm (round A) — and the borrower's terms follow from the borrower's
// two sums and its own `saved` rows.  So a member additionally publishes its planes' sums as point-readable granules (gran_p),
// and lane s fetches, for plane s of its wave, the rows of the borrower (known for the whole launch: the permutation is per
// batch) together with the item's own rows, two items ahead; after the gather it reads the borrower's granule — published
// before the borrower's member published its partial, which the gather has just seen —, re-derives the borrower's dt and
// runs bwd_plane a second time.  ONE exchange round, one workgroup barrier per item, no staged rows.  Same algebra functions
// as every other strategy (cnsn_algebra.h); the batch sums are merged in another order than the general kernels do it.
// BOXED (round 4, CN only): crop boxes (models/cnsn.py:64-82).  Four sums per plane (inside the content box; the whole plane, from
// which the outside follows by subtraction), a piecewise-affine dx with a third term inside the style box (11 coefficients),
// membership of an element from one bit mask per lane and register slot (the same for every plane).
#ifndef CNSN_SNXCN_EARLY
#define CNSN_SNXCN_EARLY 1
#endif
template <typename T, int VEC, int NV, int PPW, bool EPI, bool CN = false, bool BOXED = false>
__global__ __launch_bounds__(kBlock, snx_bwd_waves(2 * PPW * NV, EPI, VEC * (int)sizeof(T))) void resident_sn_bwd_kernel(
    std::conditional_t<CN, SnxBwdKargsCn<T>, SnxBwdKargs<T>>) {
    static_assert(!(CN && EPI), "CrossNorm with the residual-block epilogue runs the general kernels");
    static_assert(CN || !BOXED, "crop boxes belong to CrossNorm");
    constexpr int NSP = BOXED ? 4 : 2;  // floats a plane publishes for its lender
    using KA = SnxBwdKargs<T>;  // (CN: the first member of the kernel's argument, at the same offsets)
    using KAC = SnxBwdKargsCn<T>;
    using St = std::conditional_t<CN, SnxBwdStateCn, SnxBwdState>;
#define KA_ (kargs_now<KA>())
    constexpr int OWN = 4 * PPW;
    constexpr int SLOTS = 2 * PPW * NV;
    constexpr int VB = VEC * (int)sizeof(T);
    constexpr int KEEP = snx_bwd_keep(SLOTS, EPI, VB, CN, BOXED), FIRST_KEEP = SLOTS - KEEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // resident for the whole kernel: the geometry, the tensor descriptors, the item bookkeeping
    const KA* ka0 = KA_;
    const int NPARK_ = __builtin_amdgcn_readfirstlane(ka0->npark), NPARK = NPARK_;  // FIRST_KEEP <= NPARK <= SLOTS
    const int N = ka0->ra.mid.N, C = ka0->ra.mid.C, K = ka0->ra.K, M = ka0->ra.M;
    const bool has_add = EPI && ka0->addend != nullptr;
    Raw<T, VEC>* park = (Raw<T, VEC>*)smem;
    float* vals = (float*)(smem + (size_t)4 * 64 * NPARK * VB);
    St* state = (St*)((char*)vals + align16((size_t)K * kSnxVals * 4));  // [2][OWN]
    int* gave_up = (int*)(state + 2 * OWN);
    int* iperm = gave_up + 4;  // (CN) [N]: plane iperm[n] borrowed the statistics of plane n
    (void)iperm;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const PlaneIo<T, VEC, NV> sg(ka0->ra, N, C, lane);
    const __amdgpu_buffer_rsrc_t t_gy = sg.tensor(ka0->gy), t_x = sg.tensor(ka0->x), t_dx = sg.tensor(ka0->dx),
                                 t_add = sg.tensor(has_add ? ka0->addend : ka0->x);
    Raw<T, VEC>* mypark = park + (size_t)wave * NPARK * 64 + lane;
    // this workgroup is member k of cluster q_ (fixed); an "item" below is the channel the cluster works on
    const int vb_ = cluster_block(ka0->ra.xcd);  // (XCD-aware numbering: the cluster's members share one L2)
    const int q_ = vb_ / K, k = vb_ - q_ * K, nq = (int)gridDim.x / K;
    auto static_channel = [&](int seq) {
        const long long ch = (long long)q_ + (long long)seq * nq;
        return ch < (long long)C ? (int)ch : kNoChan;
    };
    // this wave's planes of ANY item: n0 .. n0 + PPW - 1, the first `nlive` of them inside the batch
    const int n0 = (k * 4 + wave) * PPW;
    const int nlive = N - n0 < 0 ? 0 : (N - n0 > PPW ? PPW : N - n0);
    const unsigned stride = (unsigned)C * (unsigned)M * (unsigned)sizeof(T);
    const int sl = lane < PPW ? lane : 0;                    // the plane of this wave the lane does the algebra for
    const int nl = n0 + sl < N ? n0 + sl : (N > 0 ? N - 1 : 0);  // (its batch index, clamped: the rows of a real plane)
#ifdef CNSN_PROF
    struct {
        unsigned long long* prof;
    } ra{ka0->ra.prof};
#endif

    if constexpr (CN) {
        const KAC* kc = kargs_now<KAC>();
        for (int n = threadIdx.x; n < N; n += kBlock) iperm[perm_at(kc->perm, kc->pin, n)] = n;
    }
    if (threadIdx.x == 0) *gave_up = 0;
    __syncthreads();
    int r_l = 0;  // (CN) lane s: the instance whose plane borrowed the statistics of plane s of this wave — the same for every item
    if constexpr (CN) r_l = iperm[nl];
    // (BOXED) bit j*VEC+q: element q of this lane's vector in register slot j lies inside the content / style box
    constexpr int MW = BOXED ? (NV * VEC + 31) / 32 : 1;
    unsigned cbits[MW], sbits[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) cbits[w] = sbits[w] = 0u;
    if constexpr (BOXED) {
        const Box cb = ka0->ra.cb, sb = ka0->ra.sb;
        const int Wd = ka0->ra.Wd;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (sg.valid(j)) {
                const int e = ((j - sg.shift) * 64 + lane) * VEC;  // VEC divides the width: one row per vector
                const int r = e / Wd, c0 = e - r * Wd;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const int p = j * VEC + q;
                    cbits[p >> 5] |= (cb.has(r, c0 + q) ? 1u : 0u) << (p & 31);
                    sbits[p >> 5] |= (sb.has(r, c0 + q) ? 1u : 0u) << (p & 31);
                }
            }
    }
    auto mask_c = [&](int j, int q) -> int {  // all-ones / all-zeros word (one v_bfe_i32)
        const int p = j * VEC + q;
        return bit_mask(cbits[p >> 5], p & 31);
    };
    auto mask_s = [&](int j, int q) -> int {
        const int p = j * VEC + q;
        return bit_mask(sbits[p >> 5], p & 31);
    };
    // the bit masks are loop-invariant, and everything derived from them would be hoisted out of the item loop — 2 x NV x VEC
    // mask words or lane-mask pairs per lane, spilled: "forgotten" at the start of every phase that uses them, so that an
    // element's mask is extracted (one v_bfe_i32) where it is used
    auto forget_masks = [&]() {
        if constexpr (BOXED) {
#pragma unroll
            for (int w = 0; w < MW; ++w) asm volatile("" : "+v"(cbits[w]), "+v"(sbits[w]));
        }
    };
    (void)mask_s;
    (void)forget_masks;
    (void)mask_c;
    startup_skew(ka0->ra);
    snx_set_priority();

    Raw<T, VEC> dg_[PPW][NV], dx_[PPW][NV];       // the item in flight
    Raw<T, VEC> da[EPI ? PPW : 1][EPI ? NV : 1];  // its addend planes
    Raw<T, VEC> keep[KEEP > 0 ? KEEP : 1];        // slots of the parked item that did not go to LDS
    double row_mu = 0.0, row_zh = 0.0, row_g = 0.0, row_sig = 0.0;  // lane s: `saved` rows of plane s of the item in flight

    auto fetch_rows = [&](int c) {  // (c: any valid channel)
        const double* saved = KA_->saved;
        const SvRec p = sv_rec(nl, c, N);
        row_mu = saved[sv_at(p, SV_MU_C)];
        row_zh = saved[sv_at(p, SV_ZH_G)];
        row_g = saved[sv_at(p, SV_G)];
        row_sig = saved[sv_at(p, SV_SIG_P)];
    };
    auto load_item = [&](int item) {
        const int c = __builtin_amdgcn_readfirstlane(item);
        fetch_rows(c);
        const unsigned span = sg.span(n0, c, C, M, true);
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const unsigned off = sg.at(span, stride, s, nlive);
#pragma unroll
            for (int j = 0; j < NV; ++j) dg_[s][j] = sg.load(t_gy, off, j);
#pragma unroll
            for (int j = 0; j < NV; ++j) dx_[s][j] = sg.load(t_x, off, j);
            if constexpr (EPI) {
                const unsigned aoff = has_add ? off : sg.dead;
#pragma unroll
                for (int j = 0; j < NV; ++j) da[s][j] = sg.load(t_add, aoff, j);
            }
        }
    };

    // per-plane sums of G' against X (shifted by the saved mean, as pass A' does), dt of each plane -> state[buf]; the
    // member's partial batch sums (round A) -> the cluster
    auto sums_publish = [&](int item, int buf) {
        const int c = __builtin_amdgcn_readfirstlane(item);
        snx_phase_fence();
        St* st = state + buf * OWN;
        // (CN) the CrossNorm rows of this lane's plane and the rows of its borrower: issued here, used behind the sums
        SnxCnRows cr{};
        if constexpr (CN) {
            const double* saved = KA_->saved;
            const SvRec p = sv_rec(nl, c, N);
            cr.mu_p = (float)saved[sv_at(p, SV_MU_P)];
            cr.aa = (float)saved[sv_at(p, SV_A)];
            cr.a1 = (float)saved[sv_at(p, SV_A1)];
            cr.m_in = (float)saved[sv_at(p, SV_M_IN)];
            cr.sig_c = (float)saved[sv_at(p, SV_SIG_C)];
            cr.M2c = (float)saved[sv_at(p, SV_M2C)];
            cr.mu_s = saved[sv_at(p, SV_MU_S)];
            cr.sig_s = (float)saved[sv_at(p, SV_SIG_S)];
            if constexpr (BOXED) cr.mu_o = (float)saved[sv_at(p, SV_MU_O)];
        }
        const int relu = EPI ? KA_->relu : 0;
        const PlaneIo<T, VEC, NV> sg(KA_->ra, 1, 1, lane);  // (only the slot validity is used here)
        const bool has_add = EPI && KA_->addend != nullptr;
        const float mu_l = (float)row_mu, g_l = (float)row_g;
        int my_s1_b = 0, my_s2_b = 0;  // lane s < PPW: the sums of this wave's plane s (v_writelane: see the forward)
        int my_s3_b = 0, my_s4_b = 0;  // (BOXED) ... outside the content box
        const float muo_l = BOXED ? cr.mu_o : 0.f;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            forget_masks();
            const float si = lane_bcast(mu_l, s);  // the saved mean of plane s, rounded as pass A' rounds it
            if constexpr (EPI) {
                if (has_add) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) dx_[s][j] = add_raw<T, VEC>(dx_[s][j], da[s][j]);
                }
                if (relu) {  // shut the gradient where the forward's output was not positive: the forward affine of a
                             // SelfNorm-only call is y = fma(float(g), X - 0, 0) whichever strategy ran it (fwd_coefs)
                    const float g = lane_bcast(g_l, s);
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
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            if constexpr (CNSN_DOT2 && sizeof(T) == 2 && !BOXED) {
                // 16 bits: sum G and sum G*X from the packed words (slots past the plane's end hold zeros), the shift by the
                // saved mean applied to the lane's two sums: sum G*(X - si) = sum G*X - si * sum G
                float gx = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int w = 0; w < VEC / 2; ++w) {
                        const unsigned gw = (unsigned)dg_[s][j][w], xw = (unsigned)dx_[s][j][w];
                        acc0 = dot2_acc<T>(gw, ones2<T>(), acc0);
                        gx = dot2_acc<T>(gw, xw, gx);
                    }
                acc1 = fmaf(-si, acc0, gx);
            } else if constexpr (BOXED) {  // content-box sums in acc0 / acc1, whole-plane sums in acc2 / acc3, both about float(mu_c)
                float acc[4];
                boxed_bwd_sums<T, VEC, NV>(dg_[s], dx_[s], [&](int j) { return sg.valid(j); }, mask_c, si, acc);
                acc0 = acc[0], acc1 = acc[1], acc2 = acc[2], acc3 = acc[3];
            } else
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
            if constexpr (BOXED) {  // outside the box = whole plane - box, re-centred on float(mu_o) (cnsn_resident_kernels.h)
                const float a0 = wave_sum(acc0), a1_ = wave_sum(acc1), a2 = wave_sum(acc2), a3 = wave_sum(acc3);
                const float so = lane_bcast(muo_l, s);
                const float o1 = a2 - a0;
                // (wave-uniform values computed on the vector unit: back to a scalar register for v_writelane)
                my_s1_b = put_lane_at(__builtin_amdgcn_readfirstlane(__float_as_int(a0)), s, my_s1_b);
                my_s2_b = put_lane_at(__builtin_amdgcn_readfirstlane(__float_as_int(a1_)), s, my_s2_b);
                my_s3_b = put_lane_at(__builtin_amdgcn_readfirstlane(__float_as_int(o1)), s, my_s3_b);
                my_s4_b = put_lane_at(__builtin_amdgcn_readfirstlane(__float_as_int((a3 - a1_) + (si - so) * o1)), s, my_s4_b);
            } else {
                my_s1_b = put_lane_at(wave_sum_bits(acc0), s, my_s1_b);
                my_s2_b = put_lane_at(wave_sum_bits(acc1), s, my_s2_b);
            }
        }
        {   // lane s < PPW: the gate's dt of plane s — one pass of arithmetic whatever PPW is; the whole record is written by
            // lane s and read back by the same lane, or by wave 0 behind the barrier (the lanes past PPW compute on zeros and
            // keep their results to themselves)
            const float my_s1 = __int_as_float(my_s1_b), my_s2 = __int_as_float(my_s2_b);
            const MidArgs a = KA_->ra.mid;
            const float my_s3 = __int_as_float(my_s3_b), my_s4 = __int_as_float(my_s4_b);
            const BwdSumsT<float> sm = fix_sums<float>(a, my_s1, my_s2, my_s3, my_s4, row_mu, (double)muo_l);
            float dtg, dtf;
            if constexpr (CN)
                gate_dt<float>(a, sm, cr.a1, cr.m_in, muo_l, cr.mu_p, g_l, 1.f, dtg, dtf);
            else
                gate_dt<float>(a, sm, 1.f, mu_l, 0.f, mu_l, g_l, 1.f, dtg, dtf);
            SnxBwdState r;
            r.mu_c = row_mu;
            r.zh = row_zh;
            r.dt = (double)dtg;
            r.g = g_l;
            r.sig_p = (float)row_sig;
            r.s1 = my_s1;
            r.s2 = my_s2;
            if constexpr (CN) {
                cr.s1o = my_s3;
                cr.s2o = my_s4;
                if (lane < PPW) st[wave * PPW + lane] = SnxBwdStateCn{r, cr};
                // this plane's sums, point-readable for the member that owns the plane which LENT it its statistics
                if (lane < PPW && n0 + lane < N) {
                    const KAC* kc = kargs_now<KAC>();
                    const unsigned epoch = kc->sn.ra.epoch;
                    const size_t pi = (size_t)c * N + (size_t)(n0 + lane);
                    if (epoch) {
                        put_tagged(kc->gran_p + NSP * pi, my_s1, epoch);
                        put_tagged(kc->gran_p + NSP * pi + 1, my_s2, epoch);
                        if constexpr (BOXED) {
                            put_tagged(kc->gran_p + NSP * pi + 2, my_s3, epoch);
                            put_tagged(kc->gran_p + NSP * pi + 3, my_s4, epoch);
                        }
                    } else {
                        put_granule(kc->gran_p + (NSP / 2) * pi, my_s1, my_s2);
                        if constexpr (BOXED) put_granule(kc->gran_p + (NSP / 2) * pi + 1, my_s3, my_s4);
                    }
                }
            } else {
                if (lane < PPW) st[wave * PPW + lane] = r;
            }
        }
        __syncthreads();
        if (wave == 0) {
            const int cnt = snx_count(N, OWN, k);
            double sa = 0.0, sb = 0.0;
            if constexpr (OWN <= 8) {  // (wave-uniform arithmetic)
                for (int i = 0; i < cnt; ++i) {
                    sa += snx_sn(st[i]).dt;
                    sb += snx_sn(st[i]).dt * snx_sn(st[i]).zh;
                }
            } else {  // lane i takes plane i of the member (OWN <= 64)
                const SnxBwdState* ri = &snx_sn(st[lane < OWN ? lane : 0]);
                const double di = lane < cnt ? ri->dt : 0.0;
                sa = wave_sum_d(di);
                sb = wave_sum_d(di * ri->zh);
            }
            const float a_hi = (float)sa, b_hi = (float)sb;
            const KA* ka = KA_;
            if (!(ka->ra.fault && c == 0 && k == K - 1))
                snx_publish(ka->gran, (size_t)c * K + k, ka->ra.epoch, a_hi, (float)(sa - (double)a_hi), b_hi,
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

    int item = static_channel(0);
    if (item == kNoChan) return;  // (the grid never exceeds the items)
    int b0 = 0;                   // state[b0]: the parked item; state[b0 ^ 1]: the item in flight
    int iter_ = 0;
    int next = static_channel(1);
    load_item(item);
    sums_publish(item, b0);
    park_item();
    if (next != kNoChan) load_item(next);

    for (;;) {
        if constexpr (snx_bwd_serial_prio(BOXED)) __builtin_amdgcn_s_setprio(3);  // the serial section ahead of the neighbours' apply loops
        const int c = __builtin_amdgcn_readfirstlane(item);
        const bool more = next != kNoChan;  // workgroup-uniform
        const int next2 = __builtin_amdgcn_readfirstlane(more ? static_channel(iter_ + 2) : kNoChan);
        const bool more2 = next2 != kNoChan;
        CNSN_STAMP(0);

        // ---- per-channel parameters of item t (scalar loads: in flight during the gather)
        float pw0, pw1, pgam;
        double prs;
        {
            const KA* ka = KA_;
            pw0 = ka->gg.w[2 * c], pw1 = ka->gg.w[2 * c + 1], pgam = ka->gg.gamma[c];
            prs = ka->saved[SV_ROWS * (size_t)N * C + c];
        }
        // (CN) lane s: the `saved` rows of the plane that borrowed the statistics of plane s of item t — in flight during the gather
        SnxBorrower br{};
        if constexpr (CN) {
            const double* saved = KA_->saved;
            const SvRec pb = sv_rec(r_l, c, N);
            br.mu_c = saved[sv_at(pb, SV_MU_C)];
            br.zh = saved[sv_at(pb, SV_ZH_G)];
            br.g = (float)saved[sv_at(pb, SV_G)];
            br.sig_p = (float)saved[sv_at(pb, SV_SIG_P)];
            br.mu_p = (float)saved[sv_at(pb, SV_MU_P)];
            br.aa = (float)saved[sv_at(pb, SV_A)];
            br.a1 = (float)saved[sv_at(pb, SV_A1)];
            br.m_in = (float)saved[sv_at(pb, SV_M_IN)];
            br.sig_c = (float)saved[sv_at(pb, SV_SIG_C)];
            br.M2c = (float)saved[sv_at(pb, SV_M2C)];
            br.mu_o = BOXED ? (float)saved[sv_at(pb, SV_MU_O)] : 0.f;
        }
        // (CN) ... and a FIRST read of the borrower's granule, also in flight during the gather: its member published it before its
        // partial, so by the time the gather has seen every partial this read has usually come back with the launch's tag — the
        // poll loop below then starts with it instead of with a memory round trip of its own (about 1 us per item on the serial
        // chain of the workgroup, which is what bounds the kernel once the outputs lie in fast-write memory)
        unsigned long long pq0 = 0, pq1 = 0, pq2 = 0, pq3 = 0;
        (void)pq0, (void)pq1, (void)pq2, (void)pq3;
        if constexpr (CN && CNSN_SNXCN_EARLY && !BOXED) {
            const KAC* kc = kargs_now<KAC>();
            const unsigned long long* gp = kc->gran_p;
            const size_t pi = (size_t)c * N + (size_t)r_l;
            if (lane < nlive) {
                if (kc->sn.ra.epoch) {
                    pq0 = __hip_atomic_load((gu64*)(gp + NSP * pi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pq1 = __hip_atomic_load((gu64*)(gp + NSP * pi + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    pq0 = __hip_atomic_load((gu64*)(gp + (NSP / 2) * pi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }

        // ---- gather round A of item t's channel
        unsigned passes_ = 0;
        {
            const KA* ka = KA_;
            const unsigned epoch = ka->ra.epoch;
            const bool got = epoch ? sweep_tagged_scalar(ka->gran + (size_t)c * K * 4, K * 4, vals, ka->ctl, ka->ra.host_flag,
                                                         ka->ra.wait_ticks, wave, epoch, passes_)
                                   : sweep_granules_scalar(ka->gran + (size_t)c * K * 2, K * 4, vals, ka->ctl, ka->ra.host_flag,
                                                           ka->ra.wait_ticks, wave, passes_);
            if (lane == 0 && !got) *gave_up = 1;
        }
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out
            T* dx = KA_->dx;
            const int first = opaque_s(n0);  // (a cold path: nothing of it is worth computing ahead of the loop)
            for (int s = 0; s < PPW; ++s) {
                const int n = first + s;
                if (n < N && lane == 0) poison_plane<T, VEC>(dx + ((size_t)n * C + c) * M);
            }
            return;
        }
        CNSN_STAMP(1);
        CNSN_NOTE(6, passes_);

        // ---- batch sums of the BatchNorm backward; dx coefficients of this wave's planes; round B
        const bool reporter = k == c % K;  // the member that writes the channel's parameter gradients
        float cG_l, cX_l, xr_l, c0_l;      // lane s: the dx coefficients of this wave's plane s of item t
        float cGo_l = 0.f, cXo_l = 0.f, xro_l = 0.f, c0o_l = 0.f, eS_l = 0.f, xs_l = 0.f, e0_l = 0.f;  // (BOXED) outside the content box; style box
        (void)cGo_l, (void)cXo_l, (void)xro_l, (void)c0o_l, (void)eS_l, (void)xs_l, (void)e0_l;
        {
            const KA* ka = KA_;
            const MidArgs a = ka->ra.mid;
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
            {   // lane s < PPW takes plane s of this wave (the others repeat plane 0); the four dx coefficients stay in
                // registers and reach the apply loop through v_readlane
                const St rec = state[b0 * OWN + wave * PPW + sl];
                const SnxBwdState r = snx_sn(rec);
                const float mu = (float)r.mu_c;
                const bool mine = lane < nlive;  // (nlive <= PPW)
                BwdSumsT<float> sm = fix_sums<float>(a, r.s1, r.s2, 0.f, 0.f, r.mu_c, 0.0);
                if constexpr (BOXED) sm = fix_sums<float>(a, r.s1, r.s2, rec.cn.s1o, rec.cn.s2o, r.mu_c, (double)rec.cn.mu_o);
                BwdPlaneT<float> o;
                BwdCoefs cf;
                float mu_p = mu;  // post-CrossNorm plane mean: SelfNorm's input statistic (the Conv1d tap gradient's factor)
                if constexpr (CN) {
                    const SnxCnRows& cr = rec.cn;
                    mu_p = cr.mu_p;
                    o = bwd_plane<float>(a, b, sm, r.dt, 0.0, r.zh, 0.0, r.g, 1.f, cr.aa, cr.a1, cr.m_in, cr.mu_p, r.sig_p, cr.sig_c,
                                         cr.M2c);
                    // the borrower's two sums: published by its member before that member's partial, which the gather has seen
                    const KAC* kc = kargs_now<KAC>();
                    const unsigned epoch = kc->sn.ra.epoch;
                    const unsigned long long* gp = kc->gran_p;
                    const size_t pi = (size_t)c * N + (size_t)r_l;
                    float s1r = 0.f, s2r = 0.f, s3r = 0.f, s4r = 0.f;
                    bool failed = false;
                    long long t_start = 0;
                    for (unsigned spins = 0;; ++spins) {
                        bool ok = true;
                        if (mine) {
                            const bool early = CNSN_SNXCN_EARLY && !BOXED && spins == 0;  // (the read sent off before the gather)
                            if (epoch) {
                                const unsigned long long q0 = early ? pq0 : __hip_atomic_load((gu64*)(gp + NSP * pi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const unsigned long long q1 = early ? pq1 : __hip_atomic_load((gu64*)(gp + NSP * pi + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ok = (unsigned)(q0 >> 32) == epoch && (unsigned)(q1 >> 32) == epoch;
                                s1r = __uint_as_float((unsigned)q0);
                                s2r = __uint_as_float((unsigned)q1);
                                if constexpr (BOXED) {
                                    const unsigned long long q2 = __hip_atomic_load((gu64*)(gp + NSP * pi + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    const unsigned long long q3 = __hip_atomic_load((gu64*)(gp + NSP * pi + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = ok && (unsigned)(q2 >> 32) == epoch && (unsigned)(q3 >> 32) == epoch;
                                    s3r = __uint_as_float((unsigned)q2);
                                    s4r = __uint_as_float((unsigned)q3);
                                }
                            } else {
                                const unsigned long long q0 = early ? pq0 : __hip_atomic_load((gu64*)(gp + (NSP / 2) * pi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ok = q0 != kGranuleEmpty;
                                s1r = __uint_as_float((unsigned)q0);
                                s2r = __uint_as_float((unsigned)(q0 >> 32));
                                if constexpr (BOXED) {
                                    const unsigned long long q1 = __hip_atomic_load((gu64*)(gp + (NSP / 2) * pi + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = ok && q1 != kGranuleEmpty;
                                    s3r = __uint_as_float((unsigned)q1);
                                    s4r = __uint_as_float((unsigned)(q1 >> 32));
                                }
                            }
                        }
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(CNSN_POLL_SLEEP);
                        if ((spins & 15u) == 15u) {
                            const long long now = (long long)wall_clock64();
                            if (t_start == 0) t_start = now;
                            unsigned* ctl = kc->sn.ctl;
                            const unsigned ctl_idle = kc->sn.ra.ctl_idle;
                            const unsigned seen = __hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (seen != ctl_idle || now - t_start > kc->sn.ra.wait_ticks) {
                                if (seen == ctl_idle && lane == 0) {
                                    const unsigned prev = __hip_atomic_exchange((gu32*)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    unsigned* host_flag = kc->sn.ra.host_flag;
                                    if (prev == ctl_idle && host_flag)
                                        __hip_atomic_fetch_add(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                }
                                failed = true;
                                break;
                            }
                        }
                    }
                    const BwdSumsT<float> smr = fix_sums<float>(a, s1r, s2r, s3r, s4r, br.mu_c, (double)br.mu_o);
                    float dtr, dtr_f;
                    gate_dt<float>(a, smr, br.a1, br.m_in, br.mu_o, br.mu_p, br.g, 1.f, dtr, dtr_f);
                    const BwdPlaneT<float> src = bwd_plane<float>(a, b, smr, (double)dtr, 0.0, br.zh, 0.0, br.g, 1.f, br.aa, br.a1,
                                                                  br.m_in, br.mu_p, br.sig_p, br.sig_c, br.M2c);
                    cf = bwd_coefs<float>(a, o, src.Emu, src.Esig, r.g, cr.a1, cr.m_in, cr.mu_p, r.mu_c, cr.sig_c, cr.mu_s, cr.sig_s);
                    if (failed) {  // the launch gives up: this wave's planes come out as NaNs, the workgroup leaves at its next gather
                        cf.c0_in = cf.c0_out = __builtin_nanf("");
                        if (lane == 0) *gave_up = 1;
                    }
                    if constexpr (BOXED) {
                        cGo_l = cf.cG_out;
                        cXo_l = cf.cX_out;
                        xro_l = cf.xr_out;
                        c0o_l = cf.c0_out;
                        eS_l = cf.eS;
                        xs_l = cf.xs;
                        e0_l = cf.e0;
                    }
                } else {
                    o = bwd_plane<float>(a, b, sm, r.dt, 0.0, r.zh, 0.0, r.g, 1.f, 1.f, 1.f, mu, mu, r.sig_p, 1.f, 0.f);
                    cf = bwd_coefs<float>(a, o, 0.f, 0.f, r.g, 1.f, mu, mu, r.mu_c, 1.f, r.mu_c, 1.f);
                }
                cG_l = cf.cG_in;
                cX_l = cf.cX_in;
                xr_l = cf.xr_in;
                c0_l = cf.c0_in;
                if constexpr (PPW == 1) {
                    pdw0 = nlive > 0 ? o.dz_g * mu_p : 0.f;  // (wave-uniform)
                    pdw1 = nlive > 0 ? o.dz_g * r.sig_p : 0.f;
                } else {
                    pdw0 = wave_sum(mine ? o.dz_g * mu_p : 0.f);
                    pdw1 = wave_sum(mine ? o.dz_g * r.sig_p : 0.f);
                }
            }
            // round B: this wave's share of the taps' gradient (lanes 0 / 1)
            const size_t wm = ((size_t)c * K + k) * 4 + wave;
            const unsigned epoch = ka->ra.epoch;
            if (epoch) {
                if (lane < 2) put_tagged(ka->gran_b + wm * 2 + lane, lane == 0 ? pdw0 : pdw1, epoch);
            } else if (lane == 0) {
                put_granule(ka->gran_b + wm, pdw0, pdw1);
            }
            if (reporter && threadIdx.x == 0) {
                ka->dgr.dgamma[c] = (float)b.s_dtz_g;
                ka->dgr.dbeta[c] = (float)b.s_dt_g;
            }
        }

        // ---- item t+1 has arrived long ago: its partial sums go out BEFORE item t is applied
        CNSN_STAMP(2);
        if (more) sums_publish(next, b0 ^ 1);
        if constexpr (snx_bwd_serial_prio(BOXED)) __builtin_amdgcn_s_setprio(0);
        CNSN_STAMP(3);

        // ---- slot by slot: apply item t (the only write of dx), park item t+1's slots in its place, send the loads of
        //      item t+2's slots after it
        {
            fetch_rows(more2 ? next2 : 0);
            // (a phase of its own for the register allocator: see the forward)
            snx_phase_fence();
            const KA* ka = KA_;
            const int N = ka->ra.mid.N, C = ka->ra.mid.C, M = ka->ra.M;
            const PlaneIo<T, VEC, NV> sg(ka->ra, N, C, lane);
            const bool has_add = EPI && ka->addend != nullptr;
            const __amdgpu_buffer_rsrc_t t_gy = sg.tensor(ka->gy), t_x = sg.tensor(ka->x), t_dx = sg.tensor(ka->dx),
                                         t_add = sg.tensor(has_add ? ka->addend : ka->x);
            const unsigned stride_ = (unsigned)C * (unsigned)M * (unsigned)sizeof(T);
            const int n0_ = opaque_s(n0);
            const int nlive_ = N - n0_ < 0 ? 0 : (N - n0_ > PPW ? PPW : N - n0_);
            const unsigned span = sg.span(n0_, c, C, M, true), span2 = sg.span(n0_, next2, C, M, more2);
            const int NPARK = __builtin_amdgcn_readfirstlane(ka->npark);  // (the parked-or-kept choices of the slots)
            auto parked = [&](int i) -> Raw<T, VEC> {
                if (i < FIRST_KEEP || i < NPARK) return mypark[i * 64];
                return keep[i - FIRST_KEEP];
            };
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                forget_masks();
                const float cG = lane_bcast(cG_l, s), cX = lane_bcast(cX_l, s), xr = lane_bcast(xr_l, s),
                            c0 = lane_bcast(c0_l, s);
                float cGo = 0.f, cXo = 0.f, xro = 0.f, c0o = 0.f, eS = 0.f, xs = 0.f, e0 = 0.f;
                float kG = cG, kX = cX, kr = xr, k0 = c0;  // (BOXED) the four in-box coefficients in VECTOR registers too
                if constexpr (BOXED) {
                    cGo = lane_bcast(cGo_l, s), cXo = lane_bcast(cXo_l, s), xro = lane_bcast(xro_l, s), c0o = lane_bcast(c0o_l, s);
                    eS = lane_bcast(eS_l, s), xs = lane_bcast(xs_l, s), e0 = lane_bcast(e0_l, s);
                    // eleven per-plane coefficients as scalar operands: a VOP3 instruction reads ONE scalar register, so the
                    // compiler copied them to vector registers per use (v_mov) and spilled the scalar file (a quarter of the
                    // kernel's vector instructions were v_readlane / v_writelane).  Pinned into vector registers once per plane.
                    asm volatile("" : "+v"(kG), "+v"(kX), "+v"(kr), "+v"(k0), "+v"(cGo), "+v"(cXo), "+v"(xro), "+v"(c0o), "+v"(eS), "+v"(xs),
                                 "+v"(e0));
                }
                const unsigned doff = sg.at(span, stride_, s, nlive_);  // (a plane past the batch end drops its stores)
                const unsigned off2 = sg.at(span2, stride_, s, nlive_);  // nothing to load: zeros, no traffic
                const unsigned aoff2 = has_add ? off2 : sg.dead;
                (void)aoff2;
                const int base = s * 2 * NV;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int ig = base + j, ix = base + NV + j;
                    const Raw<T, VEC> rg = parked(ig), rx = parked(ix);
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = elem<T, VEC>(rg, q), X = elem<T, VEC>(rx, q);
                        if constexpr (!BOXED) {
                            ov[q] = fmaf(cG, G, fmaf(cX, X - xr, c0));
                        } else if ((q & 1) == 0) {  // the three affine maps on the element PAIR (v_pk_fma_f32), then each element's by
                                                    // its mask words (no branches)
                            const cnsn_f2_t G2 = {G, elem<T, VEC>(rg, q + 1)}, X2 = {X, elem<T, VEC>(rx, q + 1)};
                            const cnsn_f2_t vi = fma2(splat2(kG), G2, fma2(splat2(kX), X2 - splat2(kr), splat2(k0)));
                            const cnsn_f2_t vo = fma2(splat2(cGo), G2, fma2(splat2(cXo), X2 - splat2(xro), splat2(c0o)));
                            const cnsn_f2_t r2 = pick_if2(mask_c(j, q), mask_c(j, q + 1), vi, vo) +
                                                 keep_if2(fma2(splat2(eS), X2 - splat2(xs), splat2(e0)), mask_s(j, q), mask_s(j, q + 1));
                            ov[q] = r2.x;
                            ov[q + 1] = r2.y;
                        }
                    }
                    sg.store(t_dx, doff, j, pack<T, VEC>(ov));
                    if (ig < FIRST_KEEP || ig < NPARK)
                        mypark[ig * 64] = dg_[s][j];
                    else
                        keep[ig - FIRST_KEEP] = dg_[s][j];
                    if (ix < FIRST_KEEP || ix < NPARK)
                        mypark[ix * 64] = dx_[s][j];
                    else
                        keep[ix - FIRST_KEEP] = dx_[s][j];
                    dg_[s][j] = sg.load(t_gy, off2, j);
                    dx_[s][j] = sg.load(t_x, off2, j);
                    if constexpr (EPI) da[s][j] = sg.load(t_add, aoff2, j);
                }
            }
            snx_phase_fence();
        }

        CNSN_STAMP(4);
        // ---- the reporter's wave 0 collects round B of channel c (4K wave shares), now that its stores are out
        if (reporter && wave == 0) {
            const KA* ka = KA_;
            const unsigned epoch = ka->ra.epoch, ctl_idle = ka->ra.ctl_idle;
            const unsigned long long* gran_b = ka->gran_b;
            unsigned* ctl = ka->ctl;
            const long long wait_ticks = ka->ra.wait_ticks;
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
                        if (epoch) {
                            const unsigned long long q0 =
                                __hip_atomic_load((gu64*)(gran_b + wm * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long q1 =
                                __hip_atomic_load((gu64*)(gran_b + wm * 2 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (unsigned)(q0 >> 32) == epoch && (unsigned)(q1 >> 32) == epoch;
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
                        if (seen != ctl_idle || now - t_start > wait_ticks) {
                            if (seen == ctl_idle && lane == 0) {
                                const unsigned prev =
                                    __hip_atomic_exchange((gu32*)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                unsigned* host_flag = ka->ra.host_flag;
                                if (prev == ctl_idle && host_flag)
                                    __hip_atomic_fetch_add(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
                float* dw = ka->dgr.dw;
                dw[2 * c] = failed ? __builtin_nanf("") : (float)s0;
                dw[2 * c + 1] = failed ? __builtin_nanf("") : (float)s1;
                if (failed) *gave_up = 1;
            }
        }
        if (!more) break;
        b0 ^= 1;
        item = next;
        next = next2;
        ++iter_;
    }
#undef KA_
}

}  // namespace cnsn
