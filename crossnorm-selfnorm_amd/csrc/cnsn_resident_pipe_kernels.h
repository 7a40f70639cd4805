// Pipelined forward of the cluster-resident strategy (round 2).
//
// The plain resident forward (cnsn_resident_kernels.h) runs load -> statistics -> publish -> cluster wait -> algebra ->
// apply -> store as ONE dependent chain per item: 18.9 us at the north-star shape, half of it the wait, with the
// memory system idle for this workgroup from its last load to its first store (profiles/r02_resident_phases.md).
// Here a workgroup works on TWO items: item t sits in LDS ("parked": its statistics are published), item t+1 is
// being loaded into registers.  Per iteration
//     gather t's channel (SCALAR memory path: a vector load would return behind the bulk loads this wave has in
//     flight)  ->  algebra of t  ->  statistics of t+1 from registers, publish  ->  slot by slot: apply t from LDS and
//     store, park t+1's slot in its place, issue the load of t+2's slot
// so the loads of an item are in flight during the whole exchange and algebra of its predecessor, the publish of t+1
// travels while t is being stored, loads and stores reach the memory system interleaved, and the chain from one publish
// to the next is gather + algebra + statistics alone.  Cost: the parked item (49 KB at 56x56 fp32) bounds residency at 3 workgroups per CU
// where the plain kernel has 4; the host side declines when LDS would not take 3.
// Same arithmetic, same published numbers, same `saved` rows as the plain kernel: results are bit-identical.
#pragma once
#include "cnsn_resident_kernels.h"
#include "cnsn_resident_io.h"

#ifndef CNSN_PIPE_PRIO
#define CNSN_PIPE_PRIO 2  // 0: none, 1: the algebra, 2: the whole serial section at wave priority 3 (profiles/r05_serial_priority.md: -1 ... -4 %)
#endif

namespace cnsn {

// ---- gather of TAGGED granules ({float, epoch} per 8 bytes) through the scalar path.  Every wave calls it
// (wave-uniform arguments); wave w owns the 256-byte groups w, w+4, ...; lane l of a group holds dword l: even lanes
// the value, odd lanes the tag.  The area is readable one group past the channel's end (context sizing).
__device__ __forceinline__ bool sweep_tagged_scalar(const unsigned long long* g, int ngran, float* vals, unsigned* ctl,
                                                    unsigned* host_flag, long long wait_ticks, int wave, unsigned epoch,
                                                    unsigned& passes) {
    const int lane = threadIdx.x & 63;
    const int ndw = 2 * ngran;
    const int ngroups = (ndw + 63) >> 6;
    const int mine = ngroups > wave ? (ngroups - wave + 3) >> 2 : 0;
    unsigned long long todo = mine >= 64 ? ~0ull : ((1ull << mine) - 1ull);
    long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool all = true;
        for (int k = 0; k < mine; ++k) {
            if (k < 64 && !((todo >> k) & 1ull)) continue;
            const int grp = wave + 4 * k;
            const unsigned v = sload_group((const void*)g, (unsigned)grp * 256u);
            const int idx = grp * 64 + lane;
            const bool valid = idx < ndw;
            if (valid && !(lane & 1)) vals[idx >> 1] = __uint_as_float(v);
            const bool ok = __ballot(valid && (lane & 1) && v != epoch) == 0ull;
            if (ok && k < 64) todo &= ~(1ull << k);
            all &= ok;
        }
        passes = spins + 1;
        if (all) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 15u) == 15u) {
            const long long now = (long long)wall_clock64();
            if (t_start == 0) t_start = now;
            if (sload_glc_u32(ctl) != 0u) return false;  // somebody gave up already (context: idle word is 0)
            if (now - t_start > wait_ticks) {
                if (lane == 0) {
                    const unsigned prev = __hip_atomic_exchange((gu32*)ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev == 0u && host_flag)
                        __hip_atomic_fetch_add(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return false;
            }
        }
    }
}

// LDS of the pipelined forward: the parked item first (16-byte aligned), then the exchange arrays
__host__ __device__ inline size_t pipe_lds_bytes(int N, int NG, int own, int parked_slots, int vec_bytes, bool cn) {
    return (size_t)4 * 64 * parked_slots * vec_bytes  // parked item: [wave][slot][lane]
           + align16((size_t)N * NG * 4)              // vals[N][NG]
           + (cn ? align16((size_t)N * 4) : 0)        // style permutation
           + align16((size_t)own * FC_ROWS * 4)       // coefficients of the owned planes
           + 4 * 4 * 8                                // block reduction scratch
           + 16;                                      // "this workgroup gave up" flag
}

// Not every slot of the item has to go to LDS: the last `SLOTS - npark` (at most kPipeKeep) stay in registers.  npark is
// a launch argument — the host parks as many slots as let three workgroups share a CU's 160 KB (at 56x56 fp32 the 13th
// slot, a quarter full, stays: 12 slots = 48 KB).
constexpr int kPipeKeep = 6;
// (one place for the host and the kernel; a knob for instantiations that spill)
constexpr int pipe_fwd_keep(int slots, int elem_bytes, bool boxed) {
    (void)elem_bytes;
    (void)boxed;
    return slots < kPipeKeep ? slots : kPipeKeep;
}
// workgroups per CU the forward is compiled for: small items (8 slots = 32 data registers) want more neighbours
constexpr int pipe_fwd_waves(int slots) { return slots <= 8 ? 4 : 3; }
// Kernel arguments travel as ONE struct and are re-read from the kernarg segment where they are used (cnsn_resident_io.h):
// the kernel has far more wave-uniform state than a wave has SGPRs (ResArgs alone is ~50 dwords), and what the compiler keeps
// resident it spills to VGPR lanes — until round 5 a fifth of this kernel's instructions were v_readlane / v_writelane.
template <typename T>
struct PipeFwdKargs {
    ResArgs ra;
    int npark;
    const T* x;
    T* y;
    const int64_t* perm;
    GateDev gg, gf;
    unsigned long long* gran;
    double* saved;
    unsigned* ctl;
    unsigned long long* clear;
    unsigned clear_n;
    PermInline pin;
};

// geometry the boxed statistics helpers ask for: slot validity from the tensor-descriptor layout, region masks from SlotGeom
template <typename IO, typename SG>
struct PipeGeom {
    const IO& io;
    const SG& sg;
    __device__ __forceinline__ bool valid(int j) const { return io.valid(j); }
    __device__ __forceinline__ int mask_c(int j, int q) const { return sg.mask_c(j, q); }
    __device__ __forceinline__ int mask_s(int j, int q) const { return sg.mask_s(j, q); }
};

template <typename T, int VEC, int NV, int PPW, bool BOXED>
__global__ __launch_bounds__(kBlock, pipe_fwd_waves(PPW * NV)) void resident_fwd_pipe_kernel(PipeFwdKargs<T>) {
    using KA = PipeFwdKargs<T>;
#define PKA_ (kargs_now<KA>())
    {
        const KA* ka = PKA_;
        pipe_clear_other_region(ka->clear, ka->clear_n);
    }
    constexpr int NG = BOXED ? 6 : 2;
    constexpr int OWN = 4 * PPW;
    constexpr int SLOTS = PPW * NV;
    constexpr int KEEP = pipe_fwd_keep(SLOTS, (int)sizeof(T), BOXED), FIRST_KEEP = SLOTS - KEEP;  // slots below FIRST_KEEP: always parked
    constexpr int VB = VEC * (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // resident for the whole kernel: the geometry, two tensor descriptors, the item bookkeeping
    const KA* ka0 = PKA_;
    const int NPARK = __builtin_amdgcn_readfirstlane(ka0->npark);  // FIRST_KEEP <= NPARK <= SLOTS
    const int N = ka0->ra.mid.N, C = ka0->ra.mid.C, K = ka0->ra.K, items = ka0->ra.items;
    const bool cn_on = ka0->ra.mid.cn_active != 0;
    Raw<T, VEC>* park = (Raw<T, VEC>*)smem;
    float* vals = (float*)(smem + (size_t)4 * 64 * NPARK * VB);
    int* sperm = (int*)((char*)vals + align16((size_t)N * NG * 4));
    float* ocoef = (float*)((char*)sperm + (cn_on ? align16((size_t)N * 4) : 0));
    double* red = (double*)((char*)ocoef + align16((size_t)OWN * FC_ROWS * 4));
    int* gave_up = (int*)(red + 4 * 4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const PlaneIo<T, VEC, NV> io(ka0->ra, N, C, lane);
    const SlotGeom<VEC, NV, BOXED> sg(ka0->ra, lane, io.shift);
    const PipeGeom<PlaneIo<T, VEC, NV>, SlotGeom<VEC, NV, BOXED>> geo{io, sg};
    const __amdgpu_buffer_rsrc_t rx = io.tensor(ka0->x), ry = io.tensor(ka0->y);
    const unsigned stride = (unsigned)C * (unsigned)ka0->ra.M * (unsigned)sizeof(T);  // plane (n, c) -> plane (n + 1, c)
    Raw<T, VEC>* mypark = park + (size_t)wave * NPARK * 64 + lane;  // slot i of this lane: mypark[i * 64]
#ifdef CNSN_PROF
    struct {
        unsigned long long* prof;
    } ra{ka0->ra.prof};
#endif

    if (cn_on) {
        const KA* ka = PKA_;
        for (int n = threadIdx.x; n < N; n += kBlock) sperm[n] = perm_at(ka->perm, ka->pin, n);
    }
    if (threadIdx.x == 0) *gave_up = 0;
    __syncthreads();
    startup_skew(ka0->ra);

    Raw<T, VEC> d[PPW][NV];                        // the item being loaded / whose statistics are being taken
    Raw<T, VEC> keep[KEEP];                        // slots of the parked item that did not go to LDS

    // this wave's planes of item (c, k): n0 .. n0 + PPW - 1, the first `nlive` of them inside the batch
    auto first_plane = [&](int k) { return (k * 4 + wave) * PPW; };
    auto live_planes = [&](int n0) { return N - n0 < 0 ? 0 : (N - n0 > PPW ? PPW : N - n0); };

    auto load_item = [&](int item, bool exists) {
        const int c = item / K, k = item - c * K;
        const int n0 = first_plane(k), nl = live_planes(n0);
        const unsigned span = io.span(n0 < N ? n0 : 0, exists ? c : 0, C, (int)(stride / ((unsigned)C * (unsigned)sizeof(T))), exists);
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const unsigned off = io.at(span, stride, s, nl);  // past the batch end / no such item: every lane reads zeros
#pragma unroll
            for (int j = 0; j < NV; ++j) d[s][j] = io.load(rx, off, j);
        }
    };

    // exact two-pass statistics of the planes in d, published to the cluster
    auto stats_publish = [&](int item) {
        snx_phase_fence();
        const KA* ka = PKA_;
        const int c = item / K, k = item - c * K;
        const int M = ka->ra.M;
        const unsigned epoch = ka->ra.epoch;
        const bool fault = ka->ra.fault && item == K - 1;
        unsigned long long* gran = ka->gran;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = first_plane(k) + s;
            float pub[NG];
            if constexpr (!BOXED) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) sum += elem<T, VEC>(d[s][j], q);
                const float mean = wave_sum(sum) / (float)M;
                float m2 = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (io.valid(j)) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            const float t = elem<T, VEC>(d[s][j], q) - mean;
                            m2 = fmaf(t, t, m2);
                        }
                    }
                pub[0] = mean;
                pub[1] = wave_sum(m2);
            } else {  // (one masked pass about the plane mean: boxed_moments, cnsn_resident_kernels.h)
                const MidArgs a = ka->ra.mid;
                const float kk = wave_sum(slots_sum<T, VEC, NV>(d[s])) / (float)a.M;  // invalid slots hold 0
                float t[6];
                boxed_region_sums<T, VEC, NV>(d[s], geo, kk, t);
#pragma unroll
                for (int m = 0; m < 6; ++m) t[m] = wave_sum(t[m]);
                boxed_moments(kk, t, a.M, a.Mc, a.Ms, pub);
            }
            if (epoch) {  // persistent context: lane m publishes pub[m] with the launch's tag
                if (n < N && lane < NG && !fault) {
                    float v = pub[0];
#pragma unroll
                    for (int m = 1; m < NG; ++m) v = (lane == m) ? pub[m] : v;
                    put_tagged(gran + ((size_t)c * N + n) * NG + lane, v, epoch);
                }
            } else if (n < N && lane < NG / 2 && !fault) {  // lane m: (pub[2m], pub[2m+1])
                float lo = pub[0], hi = pub[1];
#pragma unroll
                for (int m = 1; m < NG / 2; ++m) {
                    lo = (lane == m) ? pub[2 * m] : lo;
                    hi = (lane == m) ? pub[2 * m + 1] : hi;
                }
                put_granule(gran + ((size_t)c * N + n) * (NG / 2) + lane, lo, hi);
            }
        }
    };
    // d -> LDS (+ keep): every lane writes (and later reads back) its own slots only — no barrier involved
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

    const int grid = (int)gridDim.x;
    int item = cluster_block(ka0->ra.xcd);
    if (item >= items) return;  // (the grid never exceeds the items)
    int iter_ = 0;
    (void)iter_;
    CNSN_STAMP(0);
    load_item(item, true);
    stats_publish(item);
    park_item();
    load_item(item + grid, item + grid < items);

    for (;;) {
        const int c = item / K, k = item - c * K;
        const int next = item + grid, next2 = next + grid;
        const bool more = next < items, more2 = next2 < items;  // workgroup-uniform
        CNSN_STAMP(1);
#if CNSN_PIPE_PRIO == 2
        __builtin_amdgcn_s_setprio(3);  // (A/B: the WHOLE serial section — gather, algebra, statistics of t+1 — ahead of the neighbours' apply loops)
#endif
        snx_phase_fence();

        // ---- per-channel parameters of item t (scalar loads: in flight during the gather)
        const KA* ka = PKA_;
        const MidArgs a = ka->ra.mid;
        const GateDev gg = ka->gg, gf = ka->gf;
        double* saved = ka->saved;
        float pw[4] = {0.f, 0.f, 0.f, 0.f}, pgam[2] = {0.f, 0.f}, pbet[2] = {0.f, 0.f}, prm[2] = {0.f, 0.f},
              prv[2] = {1.f, 1.f};
        if (a.sn_active) {
            pw[0] = gg.w[2 * c];
            pw[1] = gg.w[2 * c + 1];
            pgam[0] = gg.gamma[c];
            pbet[0] = gg.beta[c];
            prm[0] = gg.run_mean[c];
            prv[0] = gg.run_var[c];
            if (a.sn_two) {
                pw[2] = gf.w[2 * c];
                pw[3] = gf.w[2 * c + 1];
                pgam[1] = gf.gamma[c];
                pbet[1] = gf.beta[c];
                prm[1] = gf.run_mean[c];
                prv[1] = gf.run_var[c];
            }
        }

        // ---- gather item t's channel.  (Round 5 tried reading the granules EARLY with a vector load issued a third of the way
        //      through the previous apply loop — counted vmcnt wait, no extra latency: profiles/r05_early_gather.md.  It misses
        //      80-100 % of the time: what this wait waits for is the slowest of the cluster's members, not a round trip.)
        unsigned passes_ = 0;
        {
            const unsigned epoch = ka->ra.epoch;
            const bool got = epoch ? sweep_tagged_scalar(ka->gran + (size_t)c * N * NG, N * NG, vals, ka->ctl, ka->ra.host_flag,
                                                         ka->ra.wait_ticks, wave, epoch, passes_)
                                   : sweep_granules_scalar(ka->gran + (size_t)c * N * (NG / 2), N * NG, vals, ka->ctl,
                                                           ka->ra.host_flag, ka->ra.wait_ticks, wave, passes_);
            if (lane == 0 && !got) *gave_up = 1;
        }
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out: see sweep_granules; the planes still owed are marked with NaNs
            T* y = PKA_->y;
            const int M = PKA_->ra.M;
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = first_plane(k) + s;
                if (n < N && lane == 0) poison_plane<T, VEC>(y + ((size_t)n * C + c) * M);
            }
            return;
        }
        CNSN_STAMP(2);
#if CNSN_PIPE_PRIO == 1
        __builtin_amdgcn_s_setprio(3);  // the algebra is the short serial section of the cycle: ahead of the neighbours' bulk loops
#endif
        CNSN_NOTE(6, passes_);

        using R = float;  // per-plane algebra in float, cross-batch sums / normalisation in double
        auto plane_of = [&](int n) {
            MomentsT<R> o;
            o.mu_c = vals[n * NG];
            o.M2c = vals[n * NG + 1];
            o.mu_o = BOXED ? vals[n * NG + 2] : 0.f;
            o.M2o = BOXED ? vals[n * NG + 3] : 0.f;
            o.mu_s = BOXED ? vals[n * NG + 4] : o.mu_c;
            o.M2s = BOXED ? vals[n * NG + 5] : o.M2c;
            R mu_sq = 0.f, M2_sq = 0.f;
            if (a.cn_active) {
                const int q = sperm[n];  // style source instance, same channel (cnsn.py:66,68)
                mu_sq = vals[q * NG + (BOXED ? 4 : 0)];
                M2_sq = vals[q * NG + (BOXED ? 5 : 1)];
            }
            return fwd_plane<R>(a, o, mu_sq, M2_sq);
        };

        // ---- SelfNorm gate statistics over the batch (every member computes them redundantly)
        const size_t P = (size_t)N * C;
        double mg = 0, mf = 0, rg = 1, rf = 1, wg0 = 0, wg1 = 0, wf0 = 0, wf1 = 0;
        if (a.sn_active) {
            wg0 = pw[0];
            wg1 = pw[1];
            wf0 = pw[2];
            wf1 = pw[3];
            if (a.sn_training) {
                const FwdPlaneT<R> f0 = plane_of(0);
                const double zs_g = wg0 * (double)f0.mu_p + wg1 * (double)f0.sig_p;
                const double zs_f = wf0 * (double)f0.mu_p + wf1 * (double)f0.sig_p;
                double sz[4] = {0.0, 0.0, 0.0, 0.0};
                for (int n = threadIdx.x; n < N; n += kBlock) {
                    const FwdPlaneT<R> f = plane_of(n);
                    const double dg = wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - zs_g;
                    const double df = wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - zs_f;
                    sz[0] += dg;
                    sz[1] += dg * dg;
                    sz[2] += df;
                    sz[3] += df * df;
                }
                block_sum_d<4>(sz, red);
                mg = zs_g + sz[0] * a.inv_n;
                mf = zs_f + sz[2] * a.inv_n;
                double vg = (sz[1] - sz[0] * sz[0] * a.inv_n) * a.inv_n, vf = (sz[3] - sz[2] * sz[2] * a.inv_n) * a.inv_n;
                vg = vg > 0.0 ? vg : 0.0;
                vf = vf > 0.0 ? vf : 0.0;
                rg = (double)__builtin_amdgcn_rsqf((float)(vg + (double)a.eps_bn));
                rf = (double)__builtin_amdgcn_rsqf((float)(vf + (double)a.eps_bn));
                if (k == 0 && threadIdx.x == 0) {
                    const double mom_ = a.momentum, unb = a.unbias_n;
                    gg.run_mean[c] = (float)((1.0 - mom_) * (double)prm[0] + mom_ * mg);
                    gg.run_var[c] = (float)((1.0 - mom_) * (double)prv[0] + mom_ * vg * unb);
                    if (a.sn_two) {
                        gf.run_mean[c] = (float)((1.0 - mom_) * (double)prm[1] + mom_ * mf);
                        gf.run_var[c] = (float)((1.0 - mom_) * (double)prv[1] + mom_ * vf * unb);
                    }
                    if (c == 0) {
                        bump_batches_tracked(gg.nbt);
                        if (a.sn_two) bump_batches_tracked(gf.nbt);
                    }
                }
            } else {
                mg = prm[0];
                rg = (double)__builtin_amdgcn_rsqf(prv[0] + a.eps_bn);
                if (a.sn_two) {
                    mf = prm[1];
                    rf = (double)__builtin_amdgcn_rsqf(prv[1] + a.eps_bn);
                }
            }
            if (saved && k == 0 && threadIdx.x == 0) {
                saved[SV_ROWS * P + c] = rg;
                saved[SV_ROWS * P + C + c] = rf;
            }
        }

        // ---- coefficients (and saved state) of the owned planes
        if (threadIdx.x < OWN) {
            const int n = k * OWN + threadIdx.x;
            if (n < N) {
                const FwdPlaneT<R> f = plane_of(n);
                R g = 1.f, fg = 1.f;
                double zhg = 0.0, zhf = 0.0;
                if (a.sn_active) {
                    zhg = (wg0 * (double)f.mu_p + wg1 * (double)f.sig_p - mg) * rg;
                    g = sigmoid_r<R>((R)((double)pgam[0] * zhg + (double)pbet[0]));
                    if (a.sn_two) {
                        zhf = (wf0 * (double)f.mu_p + wf1 * (double)f.sig_p - mf) * rf;
                        fg = sigmoid_r<R>((R)((double)pgam[1] * zhf + (double)pbet[1]));
                    }
                }
                const FwdCoefs cf = fwd_coefs<R>(a, f, g, fg);
                float* o = ocoef + threadIdx.x * FC_ROWS;
                o[FC_A_IN] = cf.a_in;
                o[FC_XR] = cf.xr;
                o[FC_B_IN] = cf.b_in;
                o[FC_A_OUT] = cf.a_out;
                o[FC_B_OUT] = cf.b_out;
                if (saved) {
                    const SvRec p = sv_rec(n, c, N);
                    store_fwd_plane<R>(saved, P, p, f, a.cn_active);
                    saved[sv_at(p, SV_G)] = g;
                    saved[sv_at(p, SV_ZH_G)] = zhg;
                    saved[sv_at(p, SV_F)] = fg;
                    saved[sv_at(p, SV_ZH_F)] = zhf;
                    if (a.save_coefs) store_fwd_coefs(saved, p, cf);
                }
            }
        }
        __syncthreads();
#if CNSN_PIPE_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        CNSN_STAMP(3);

        // ---- item t+1 has arrived long ago: its statistics go out BEFORE item t is applied — the cluster's next
        //      exchange travels while this workgroup stores
        if (more) stats_publish(next);
#if CNSN_PIPE_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        CNSN_STAMP(4);
        snx_phase_fence();

        // ---- slot by slot: apply item t from LDS (the only write of y), park item t+1's slot in its place, send the
        //      load of item t+2's slot (the only read of x) after it: item t's stores and item t+2's loads reach the
        //      memory system interleaved, and the loads are under way a whole apply phase earlier
        {
            const int c2 = next2 / K, k2 = next2 - c2 * K;
            const int M_ = (int)(stride / ((unsigned)C * (unsigned)sizeof(T)));
            const int n0 = first_plane(k), n02 = first_plane(k2);
            const unsigned span_y = io.span(n0 < N ? n0 : 0, c, C, M_, true);
            const unsigned span_x2 = io.span(n02 < N ? n02 : 0, more2 ? c2 : 0, C, M_, more2);
            const int nl = live_planes(n0), nl2 = live_planes(n02);
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                sg.forget();
                const float* o = ocoef + (wave * PPW + s) * FC_ROWS;
                const float a_in = o[FC_A_IN], xr = o[FC_XR], b_in = o[FC_B_IN], a_out = o[FC_A_OUT], b_out = o[FC_B_OUT];
                const unsigned off_y = io.at(span_y, stride, s, nl);      // (a plane past the batch end drops its stores)
                const unsigned off_x2 = io.at(span_x2, stride, s, nl2);   // nothing to load: zeros, no traffic
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
                        const float f = elem<T, VEC>(v, q);
                        if constexpr (!BOXED) {
                            ov[q] = fmaf(a_in, f - xr, b_in);
                        } else if ((q & 1) == 0) {  // both maps on the element PAIR, each element's picked by its mask word
                            const cnsn_f2_t f2 = {f, elem<T, VEC>(v, q + 1)};
                            const cnsn_f2_t r2 = pick_if2(sg.mask_c(j, q), sg.mask_c(j, q + 1), fma2(splat2(a_in), f2 - splat2(xr), splat2(b_in)),
                                                          fma2(splat2(a_out), f2, splat2(b_out)));
                            ov[q] = r2.x;
                            ov[q + 1] = r2.y;
                        }
                    }
                    io.store(ry, off_y, j, pack<T, VEC>(ov));
                    if (i < FIRST_KEEP || i < NPARK)  // park slot i of item t+1 (garbage after the last item: never read)
                        mypark[i * 64] = d[s][j];
                    else
                        keep[i - FIRST_KEEP] = d[s][j];
                    d[s][j] = io.load(rx, off_x2, j);
                }
            }
        }
        if (!more) break;
        CNSN_STAMP(5);
        item = next;
        ++iter_;
    }
#undef PKA_
}

// ================================================================================================
// pipelined backward
// ================================================================================================
// The same schedule for the backward: item t (its G and x planes, 2 * NV slots per plane) waits parked while item t+1
// loads.  An item is twice the forward's, so residency is TWO workgroups per CU (256 VGPRs per lane): the first `npark`
// slots of the parked item go to LDS (80 KB per workgroup less the staged `saved` rows: 13 slots at N = 256), the rest
// stays in registers next to the item being loaded.  Slot order: G slots first (s * 2 * NV + j), then x slots
// (s * 2 * NV + NV + j).
__host__ __device__ inline size_t pipe_bwd_lds_bytes(int N, int NS, int own, int parked_slots, int vec_bytes) {
    return (size_t)4 * 64 * parked_slots * vec_bytes  // parked slots: [wave][slot][lane]
           + align16((size_t)N * NS * 4)              // vals[N][NS]
           + align16((size_t)2 * N * 8)               // dt [2][N]
           + align16((size_t)N * 4)                   // inverse permutation
           + align16((size_t)own * BC_ROWS * 4)       // coefficients of the owned planes
           + 4 * 4 * 8 + 16                           // block reduction scratch, "gave up" flag
           + align16((size_t)N * D_N * 8) + align16((size_t)N * F_N * 4);  // staged `saved` rows
}

// slots below this index always go to LDS; workgroups per CU the backward is compiled for (items of up to 16 slots:
// three, with fewer slots in LDS and more in registers)
constexpr int pipe_bwd_first_keep(int slots) { return slots <= 16 ? 6 : 12; }
constexpr int pipe_bwd_waves(int slots) { return slots <= 16 ? 3 : 2; }
template <typename T, int VEC, int NV, int PPW, bool BOXED>
__global__ __launch_bounds__(kBlock, pipe_bwd_waves(2 * PPW * NV)) void resident_bwd_pipe_kernel(ResArgs ra, int npark, const T* __restrict__ gy,
                                                                      const T* __restrict__ x, T* __restrict__ dx,
                                                                      const int64_t* __restrict__ perm, GateDev gg, GateDev gf,
                                                                      GateGradDev dgr, GateGradDev dfr,
                                                                      unsigned long long* __restrict__ gran,
                                                                      const double* __restrict__ saved,
                                                                      unsigned* __restrict__ ctl,
                                                                      unsigned long long* __restrict__ clear, unsigned clear_n,
                                                                      PermInline pin) {
    pipe_clear_other_region(clear, clear_n);
    constexpr int NS = BOXED ? 4 : 2;
    constexpr int OWN = 4 * PPW;
    constexpr int SLOTS = 2 * PPW * NV;
    constexpr int FIRST_KEEP = pipe_bwd_first_keep(SLOTS), KEEP = SLOTS - FIRST_KEEP;
    constexpr int VB = VEC * (int)sizeof(T);
    const int NPARK = __builtin_amdgcn_readfirstlane(npark);  // FIRST_KEEP <= NPARK <= SLOTS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MidArgs a = ra.mid;
    const int N = a.N, C = a.C;
    Raw<T, VEC>* park = (Raw<T, VEC>*)smem;
    float* vals = (float*)(smem + (size_t)4 * 64 * NPARK * VB);
    double* dtb = (double*)((char*)vals + align16((size_t)N * NS * 4));
    int* iperm = (int*)((char*)dtb + align16((size_t)2 * N * 8));
    float* ocoef = (float*)((char*)iperm + align16((size_t)N * 4));
    double* red = (double*)((char*)ocoef + align16((size_t)OWN * BC_ROWS * 4));
    int* gave_up = (int*)(red + 4 * 4);
    double* svd = red + 4 * 4 + 2;                                      // [N][D_N]
    float* svf = (float*)((char*)svd + align16((size_t)N * D_N * 8));  // [N][F_N]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t P = (size_t)N * C;
    const SlotGeom<VEC, NV, BOXED> sg(ra, lane);
    const int voff = lane * VB;
    Raw<T, VEC>* mypark = park + (size_t)wave * NPARK * 64 + lane;  // slot i of this lane: mypark[i * 64]

    if (a.cn_active)  // plane r receives the style-statistic gradient of the plane that borrowed from it
        for (int n = threadIdx.x; n < N; n += kBlock) iperm[perm_at(perm, pin, n)] = n;
    if (threadIdx.x == 0) *gave_up = 0;
    __syncthreads();
    startup_skew(ra);

    Raw<T, VEC> dg_[PPW][NV], dx_[PPW][NV];  // the item being loaded / whose sums are being taken
    Raw<T, VEC> keep[KEEP > 0 ? KEEP : 1];   // slots of the parked item that did not go to LDS
    float own_si[PPW], own_so[PPW];          // saved means of the planes in dg_/dx_ (the shift of their sums)

    auto load_item = [&](int item) {
        const int c = item / ra.K, k = item - c * ra.K;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const int n = (k * 4 + wave) * PPW + s;
            const SvRec p = sv_rec((n < N ? n : 0), c, N);
            own_si[s] = (float)saved[sv_at(p, SV_MU_C)];
            own_so[s] = BOXED ? (float)saved[sv_at(p, SV_MU_O)] : 0.f;
            const size_t off = ((size_t)(n < N ? n : 0) * C + c) * ra.M;
            const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) dg_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(gy + off, pbytes, j), voff);
#pragma unroll
            for (int j = 0; j < NV; ++j) dx_[s][j] = buf_load<T, VEC>(slot_rsrc<T, VEC>(x + off, pbytes, j), voff);
        }
    };

    // per-plane sums of G against x (shifted by the saved means, as pass A' does), published to the cluster
    auto sums_publish = [&](int item) {
        const int c = item / ra.K, k = item - c * ra.K;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            sg.forget();
            const int n = (k * 4 + wave) * PPW + s;
            const float si = own_si[s], so = own_so[s];
            float acc[NS];
#pragma unroll
            for (int m = 0; m < NS; ++m) acc[m] = 0.f;
            if constexpr (BOXED) {  // whole-plane sums in acc[2..3], content-box sums in acc[0..1], both about float(mu_c)
                boxed_bwd_sums<T, VEC, NV>(dg_[s], dx_[s], [&](int j) { return sg.valid(j); }, [&](int j, int q) { return sg.mask_c(j, q); }, si, acc);
            } else
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (sg.valid(j)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = elem<T, VEC>(dg_[s][j], q), X = elem<T, VEC>(dx_[s][j], q);
                        acc[0] += G;
                        acc[1] = fmaf(G, X - si, acc[1]);
                    }
                }
#pragma unroll
            for (int m = 0; m < NS; ++m) acc[m] = wave_sum(acc[m]);
            if constexpr (BOXED) {  // outside the box = whole plane - box, re-centred on float(mu_o) (cnsn_resident_kernels.h)
                const float o1 = acc[2] - acc[0];
                acc[3] = (acc[3] - acc[1]) + (si - so) * o1;
                acc[2] = o1;
            }
            if (ra.epoch) {
                if (n < N && lane < NS && !(ra.fault && item == ra.K - 1)) {
                    float v = acc[0];
#pragma unroll
                    for (int m = 1; m < NS; ++m) v = (lane == m) ? acc[m] : v;
                    put_tagged(gran + ((size_t)c * N + n) * NS + lane, v, ra.epoch);
                }
            } else if (n < N && lane < NS / 2 && !(ra.fault && item == ra.K - 1)) {
                float lo = acc[0], hi = acc[1];
#pragma unroll
                for (int m = 1; m < NS / 2; ++m) {
                    lo = (lane == m) ? acc[2 * m] : lo;
                    hi = (lane == m) ? acc[2 * m + 1] : hi;
                }
                put_granule(gran + ((size_t)c * N + n) * (NS / 2) + lane, lo, hi);
            }
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

    // `saved` rows of a whole channel for the algebra, in two halves: fetch (global -> registers, instance threadIdx.x)
    // and stage (registers -> LDS).  The fetch of item t+1 is issued before item t's stores and staged after them, so
    // its latency is nobody's wait.  Instances past the first 256 (N > 256) are fetched and staged in one go.
    struct Rows {
        double mu_c, zh_g, zh_f, mu_s;
        float a1, m_in, mu_o, mu_p, g, f, aa, sig_p, sig_c, M2c, sig_s;
    };
    auto rows_of = [&](int c, int n) {
        const SvRec p = sv_rec(n, c, N);
        Rows r;
        r.mu_c = saved[sv_at(p, SV_MU_C)];
        r.zh_g = saved[sv_at(p, SV_ZH_G)];
        r.zh_f = saved[sv_at(p, SV_ZH_F)];
        const CnRowsT<float> cr = load_cn_rows<float>(a, saved, p, r.mu_c);
        r.mu_s = cr.mu_s;
        r.a1 = cr.a1;
        r.m_in = cr.m_in;
        r.mu_o = cr.mu_o;
        r.mu_p = (float)saved[sv_at(p, SV_MU_P)];
        r.g = (float)saved[sv_at(p, SV_G)];
        r.f = (float)saved[sv_at(p, SV_F)];
        r.aa = cr.aa;
        r.sig_p = (float)saved[sv_at(p, SV_SIG_P)];
        r.sig_c = cr.sig_c;
        r.M2c = cr.M2c;
        r.sig_s = cr.sig_s;
        return r;
    };
    auto stage_row = [&](int n, const Rows& r) {
        float* sf = svf + n * F_N;
        double* sd = svd + n * D_N;
        sd[D_MU_C] = r.mu_c;
        sd[D_ZH_G] = r.zh_g;
        sd[D_ZH_F] = r.zh_f;
        sd[D_MU_S] = r.mu_s;
        sf[F_A1] = r.a1;
        sf[F_M_IN] = r.m_in;
        sf[F_MU_O] = r.mu_o;
        sf[F_MU_P] = r.mu_p;
        sf[F_G] = r.g;
        sf[F_F] = r.f;
        sf[F_A] = r.aa;
        sf[F_SIG_P] = r.sig_p;
        sf[F_SIG_C] = r.sig_c;
        sf[F_M2C] = r.M2c;
        sf[F_SIG_S] = r.sig_s;
    };
    auto stage_rows = [&](int c, const Rows& mine) {
        if ((int)threadIdx.x < N) stage_row(threadIdx.x, mine);
        for (int n = threadIdx.x + kBlock; n < N; n += kBlock) stage_row(n, rows_of(c, n));
    };

    int item = cluster_block(ra.xcd);
    if (item >= ra.items) return;
    int iter_ = 0;
    (void)iter_;
    CNSN_STAMP(0);
    load_item(item);
    {
        const int c0 = item / ra.K;
        stage_rows(c0, rows_of(c0, (int)threadIdx.x < N ? threadIdx.x : 0));
    }
    sums_publish(item);
    park_item();
    if (item + (int)gridDim.x < ra.items) load_item(item + (int)gridDim.x);

    for (;;) {
        const int c = item / ra.K, k = item - c * ra.K;
        const int next = item + (int)gridDim.x, next2 = next + (int)gridDim.x;
        const bool more = next < ra.items, more2 = next2 < ra.items;  // workgroup-uniform
        CNSN_STAMP(1);

        // ---- per-channel parameters of item t (scalar loads; its `saved` rows were staged at the end of the previous
        //      iteration; item t+1's planes have been on their way into registers since then, too)
        float pw[4] = {0.f, 0.f, 0.f, 0.f}, pgam[2] = {0.f, 0.f};
        double prs[2] = {1.0, 1.0};
        if (a.sn_active) {
            pw[0] = gg.w[2 * c];
            pw[1] = gg.w[2 * c + 1];
            pgam[0] = gg.gamma[c];
            prs[0] = saved[SV_ROWS * P + c];
            if (a.sn_two) {
                pw[2] = gf.w[2 * c];
                pw[3] = gf.w[2 * c + 1];
                pgam[1] = gf.gamma[c];
                prs[1] = saved[SV_ROWS * P + C + c];
            }
        }

        // ---- gather item t's channel (scalar path)
        unsigned passes_ = 0;
        {
            const bool got = ra.epoch ? sweep_tagged_scalar(gran + (size_t)c * N * NS, N * NS, vals, ctl, ra.host_flag,
                                                            ra.wait_ticks, wave, ra.epoch, passes_)
                                      : sweep_granules_scalar(gran + (size_t)c * N * (NS / 2), N * NS, vals, ctl, ra.host_flag,
                                                              ra.wait_ticks, wave, passes_);
            if (lane == 0 && !got) *gave_up = 1;
        }
        __syncthreads();
        if (*gave_up) {  // (workgroup-uniform) timed out: see sweep_granules; the planes still owed are marked with NaNs
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                const int n = (k * 4 + wave) * PPW + s;
                if (n < N && lane == 0) poison_plane<T, VEC>(dx + ((size_t)n * C + c) * ra.M);
            }
            return;
        }
        CNSN_STAMP(2);
#if CNSN_PIPE_PRIO
        __builtin_amdgcn_s_setprio(3);  // the algebra is the short serial section of the cycle: ahead of the neighbours' bulk loops
#endif
        CNSN_NOTE(6, passes_);

        using R = float;  // per-plane algebra in float; batch sums and the dz line in double
        auto sums_of = [&](int n) {
            return fix_sums<R>(a, vals[n * NS], vals[n * NS + 1], BOXED ? vals[n * NS + 2] : 0.f,
                               BOXED ? vals[n * NS + 3] : 0.f, svd[n * D_N + D_MU_C], (double)svf[n * F_N + F_MU_O]);
        };

        // ---- gate backward: dt for every instance of the channel, batch sums (all members)
        double s4[4] = {0, 0, 0, 0};
        BnBwd b{};
        if (a.sn_active) {
            for (int n = threadIdx.x; n < N; n += kBlock) {
                const float* sf = svf + n * F_N;
                R dtg, dtf;
                gate_dt<R>(a, sums_of(n), sf[F_A1], sf[F_M_IN], sf[F_MU_O], sf[F_MU_P], sf[F_G], sf[F_F], dtg, dtf);
                s4[0] += (double)dtg;
                s4[1] += (double)dtg * svd[n * D_N + D_ZH_G];
                s4[2] += (double)dtf;
                s4[3] += (double)dtf * svd[n * D_N + D_ZH_F];
                dtb[n] = dtg;
                dtb[N + n] = dtf;
            }
            block_sum_d<4>(s4, red);
            b.s_dt_g = s4[0];
            b.s_dtz_g = s4[1];
            b.s_dt_f = s4[2];
            b.s_dtz_f = s4[3];
            b.wg0 = pw[0];
            b.wg1 = pw[1];
            b.kg = (double)pgam[0] * prs[0];
            b.wf0 = pw[2];
            b.wf1 = pw[3];
            b.kf = (double)pgam[1] * prs[1];
        }

        auto bwd_of = [&](int n) {
            const float* sf = svf + n * F_N;
            return bwd_plane<R>(a, b, sums_of(n), a.sn_active ? dtb[n] : 0.0, a.sn_active ? dtb[N + n] : 0.0,
                                svd[n * D_N + D_ZH_G], svd[n * D_N + D_ZH_F], sf[F_G], sf[F_F], sf[F_A], sf[F_A1],
                                sf[F_M_IN], sf[F_MU_P], sf[F_SIG_P], sf[F_SIG_C], sf[F_M2C]);
        };

        // ---- parameter gradients of the channel: one member per channel (rotating) does the sums
        if (a.sn_active && k == c % ra.K) {
            double sw[4] = {0, 0, 0, 0};
            for (int n = threadIdx.x; n < N; n += kBlock) {
                const BwdPlaneT<R> o = bwd_of(n);
                const double mu_p = svf[n * F_N + F_MU_P], sig_p = svf[n * F_N + F_SIG_P];
                sw[0] += (double)o.dz_g * mu_p;
                sw[1] += (double)o.dz_g * sig_p;
                sw[2] += (double)o.dz_f * mu_p;
                sw[3] += (double)o.dz_f * sig_p;
            }
            block_sum_d<4>(sw, red);
            if (threadIdx.x == 0) {
                dgr.dgamma[c] = (float)s4[1];
                dgr.dbeta[c] = (float)s4[0];
                dgr.dw[2 * c] = (float)sw[0];
                dgr.dw[2 * c + 1] = (float)sw[1];
                if (a.sn_two) {
                    dfr.dgamma[c] = (float)s4[3];
                    dfr.dbeta[c] = (float)s4[2];
                    dfr.dw[2 * c] = (float)sw[2];
                    dfr.dw[2 * c + 1] = (float)sw[3];
                }
            }
        }

        // ---- coefficients of dx for the owned planes
        if (threadIdx.x < OWN) {
            const int n = k * OWN + threadIdx.x;
            if (n < N) {
                const float* sf = svf + n * F_N;
                const BwdPlaneT<R> o = bwd_of(n);
                R Emu = 0.f, Esig = 0.f;
                if (a.cn_active) {
                    const BwdPlaneT<R> src = bwd_of(iperm[n]);  // the plane that used (n,c) as its style
                    Emu = src.Emu;
                    Esig = src.Esig;
                }
                const BwdCoefs cf =
                    bwd_coefs<R>(a, o, Emu, Esig, sf[F_G], sf[F_A1], sf[F_M_IN], sf[F_MU_P], svd[n * D_N + D_MU_C],
                                 sf[F_SIG_C], svd[n * D_N + D_MU_S], sf[F_SIG_S]);
                float* oc = ocoef + threadIdx.x * BC_ROWS;
                oc[BC_CG_IN] = cf.cG_in;
                oc[BC_CX_IN] = cf.cX_in;
                oc[BC_XR_IN] = cf.xr_in;
                oc[BC_C0_IN] = cf.c0_in;
                oc[BC_CG_OUT] = cf.cG_out;
                oc[BC_CX_OUT] = cf.cX_out;
                oc[BC_XR_OUT] = cf.xr_out;
                oc[BC_C0_OUT] = cf.c0_out;
                oc[BC_ES] = cf.eS;
                oc[BC_XS] = cf.xs;
                oc[BC_E0] = cf.e0;
            }
        }
        __syncthreads();  // (also: every wave is done reading svd / svf before the next iteration restages them)
#if CNSN_PIPE_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        CNSN_STAMP(3);

        // ---- item t+1 has arrived long ago: its sums go out BEFORE item t is applied; the fetch of its channel's
        //      `saved` rows is issued ahead of item t's stores
        const int cn_ = next / ra.K;
        Rows nrow{};
        if (more) {
            sums_publish(next);
            nrow = rows_of(cn_, (int)threadIdx.x < N ? threadIdx.x : 0);
        }
        CNSN_STAMP(4);

        // ---- slot by slot: apply item t from LDS / the kept registers (the only write of dx), park item t+1's slot in
        //      its place, and send the load of item t+2's slot after it — item t's stores and item t+2's loads reach
        //      the memory system interleaved, and the loads are under way a whole apply phase earlier than they would
        //      be after it
        {
            const int c2 = next2 / ra.K, k2 = next2 - c2 * ra.K;
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                sg.forget();
                const int n = (k * 4 + wave) * PPW + s;
                const float* oc = ocoef + (wave * PPW + s) * BC_ROWS;
                const float cG_i = oc[BC_CG_IN], cX_i = oc[BC_CX_IN], xr_i = oc[BC_XR_IN], c0_i = oc[BC_C0_IN];
                const float cG_o = oc[BC_CG_OUT], cX_o = oc[BC_CX_OUT], xr_o = oc[BC_XR_OUT], c0_o = oc[BC_C0_OUT];
                const float eS = oc[BC_ES], xs = oc[BC_XS], e0 = oc[BC_E0];
                T* db = dx + ((size_t)(n < N ? n : 0) * C + c) * ra.M;
                const int pbytes = n < N ? ra.M * (int)sizeof(T) : 0;  // (a plane past the batch end drops its stores)
                // item t+2 (this wave's plane s of it)
                const int n2 = (k2 * 4 + wave) * PPW + s;
                const bool live2 = more2 && n2 < N;
                const SvRec p2 = sv_rec(live2 ? n2 : 0, more2 ? c2 : 0, N);
                const float si2 = (float)saved[sv_at(p2, SV_MU_C)];
                const float so2 = BOXED ? (float)saved[sv_at(p2, SV_MU_O)] : 0.f;
                const size_t off2 = ((size_t)(live2 ? n2 : 0) * C + (more2 ? c2 : 0)) * ra.M;
                const int pbytes2 = live2 ? ra.M * (int)sizeof(T) : 0;  // nothing to load: every lane reads zeros, no traffic
                const int base = s * 2 * NV;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int ig = base + j, ix = base + NV + j;
                    const Raw<T, VEC> rg = parked(ig), rx = parked(ix);
                    float ov[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float G = elem<T, VEC>(rg, q), X = elem<T, VEC>(rx, q);
                        float v;
                        if constexpr (!BOXED) {
                            v = fmaf(cG_i, G, fmaf(cX_i, X - xr_i, c0_i));
                        } else {
                            if ((q & 1) == 0) {  // the three affine maps on the element PAIR, then each element's by its mask words
                                const cnsn_f2_t G2 = {G, elem<T, VEC>(rg, q + 1)}, X2 = {X, elem<T, VEC>(rx, q + 1)};
                                const cnsn_f2_t vi = fma2(splat2(cG_i), G2, fma2(splat2(cX_i), X2 - splat2(xr_i), splat2(c0_i)));
                                const cnsn_f2_t vo = fma2(splat2(cG_o), G2, fma2(splat2(cX_o), X2 - splat2(xr_o), splat2(c0_o)));
                                const cnsn_f2_t r2 = pick_if2(sg.mask_c(j, q), sg.mask_c(j, q + 1), vi, vo) +
                                                     keep_if2(fma2(splat2(eS), X2 - splat2(xs), splat2(e0)), sg.mask_s(j, q), sg.mask_s(j, q + 1));
                                ov[q] = r2.x;
                                ov[q + 1] = r2.y;
                            }
                            continue;
                        }
                        ov[q] = v;
                    }
                    buf_store<T, VEC>(slot_rsrc<T, VEC>(db, pbytes, j), voff, pack<T, VEC>(ov));
                    // park slot j of item t+1 (garbage after the last item: never read)
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
                }
                own_si[s] = si2;
                own_so[s] = so2;
            }
        }
        if (!more) break;
        stage_rows(cn_, nrow);  // (every wave left the algebra of item t at the barrier above)
        CNSN_STAMP(5);
        item = next;
        ++iter_;
    }
}

}  // namespace cnsn
