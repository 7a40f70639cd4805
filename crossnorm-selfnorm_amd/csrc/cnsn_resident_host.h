// Host side of the channel-resident strategy, shared by its two translation units (cnsn_resident.hip: the op
// alone; cnsn_resident_fused.hip: the op with the residual-block epilogue): eligibility, launch geometry, dispatch.
#pragma once
#include <hip/hip_runtime.h>
#include <optional>
#include "cnsn_env.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "cnsn_host_common.h"
#include "cnsn_resident_kernels.h"

namespace cnsn {
namespace reshost {


// vectors per lane and plane (register bucket) -> planes per wave.  Chosen so a wave keeps <= 16 vectors
// (64 VGPRs) per tensor in flight forward, twice that backward (G and x).
//   - the backward of the 7/8-slot buckets takes two planes per wave: measured 0.283 vs 0.355 ms on the
//     (256,256,56,56) bf16 backward; the forward is faster with one;
//   - with the residual-block epilogue a second (forward) / third (backward) tensor is in flight: fewer planes
//     per wave wherever the compiler would otherwise spill.
constexpr int kBucketNv[] = {1, 2, 4, 7, 8, 13, 16};
#ifndef CNSN_PPW78_F32
#define CNSN_PPW78_F32 2
#endif
// (16-bit, 8 slots, TWO planes per wave, boxed: hipcc of ROCm 7.2 produces wrong gradients for that instantiation at
//  168 VGPRs + scratch, and right ones when given 256 VGPRs — a spill-path miscompile; caught by
//  tests/test_gpu_resident_instantiations.py.  The 8-slot class therefore holds one plane per wave, which is also
//  faster: 0.087 vs 0.094 ms on the (64,256,64,64) bf16 backward)
#ifndef CNSN_PPW8_BWD16
#define CNSN_PPW8_BWD16 1
#endif
#ifndef CNSN_PPW78_FWD16
#define CNSN_PPW78_FWD16 2
#endif
#ifndef CNSN_PPW1
#define CNSN_PPW1 8
#endif
#ifndef CNSN_PPW4
#define CNSN_PPW4 4
#endif
#ifndef CNSN_PPW4_EPI
#define CNSN_PPW4_EPI 2
#endif
#ifndef CNSN_PPW78_EPI16_BWD
#define CNSN_PPW78_EPI16_BWD 1
#endif
#ifndef CNSN_PPW78_EPI16_FWD
#define CNSN_PPW78_EPI16_FWD 1
#endif
#ifndef CNSN_PPW2_FWD
#define CNSN_PPW2_FWD 4
#endif
#ifndef CNSN_PPW2_BWD
#define CNSN_PPW2_BWD 4
#endif
constexpr int ppw_of(int nv, bool backward, bool epi, int elem_bytes) {
    return nv == 1 ? CNSN_PPW1
           : nv == 2 ? (backward ? CNSN_PPW2_BWD : CNSN_PPW2_FWD)
           : nv == 4 ? (epi ? CNSN_PPW4_EPI : CNSN_PPW4)
           : (nv == 7 || nv == 8) ? ((backward && !epi) ? (elem_bytes == 4 ? CNSN_PPW78_F32 : (nv == 8 ? CNSN_PPW8_BWD16 : 2))
                                                     : (elem_bytes == 2 ? (epi ? (backward ? CNSN_PPW78_EPI16_BWD : CNSN_PPW78_EPI16_FWD)
                                                                               : CNSN_PPW78_FWD16)
                                                                        : 1))
                                  : 1;
}

static inline int cu_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

template <bool BWD, bool EPI, typename F>
bool dispatch_res(int dtype, int vec, int nv, F&& f) {
    auto by_nv = [&](auto tt, auto vt) -> bool {
        switch (nv) {
            case 1: f(tt, vt, IntTag<1>{}, IntTag<ppw_of(1, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            case 2: f(tt, vt, IntTag<2>{}, IntTag<ppw_of(2, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            case 4: f(tt, vt, IntTag<4>{}, IntTag<ppw_of(4, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            case 7: f(tt, vt, IntTag<7>{}, IntTag<ppw_of(7, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            case 8: f(tt, vt, IntTag<8>{}, IntTag<ppw_of(8, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            case 13: f(tt, vt, IntTag<13>{}, IntTag<ppw_of(13, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            case 16: f(tt, vt, IntTag<16>{}, IntTag<ppw_of(16, BWD, EPI, (int)sizeof(typename decltype(tt)::type))>{}); return true;
            default: return false;
        }
    };
    if (dtype == CNSN_F32 && vec == 4) return by_nv(TypeTag<float>{}, IntTag<4>{});
    if (dtype == CNSN_BF16 && vec == 8) return by_nv(TypeTag<bf16_t>{}, IntTag<8>{});
    if (dtype == CNSN_BF16 && vec == 4) return by_nv(TypeTag<bf16_t>{}, IntTag<4>{});
    if (dtype == CNSN_F16 && vec == 8) return by_nv(TypeTag<_Float16>{}, IntTag<8>{});
    if (dtype == CNSN_F16 && vec == 4) return by_nv(TypeTag<_Float16>{}, IntTag<4>{});
    return false;
}

static inline ResArgs make_args(const cnsn_problem_t& p, Box cb, Box sb, const MidArgs& mid, const ResPlan& rp) {
    ResArgs ra;
    ra.mid = mid;
    ra.M = p.H * p.W;
    ra.Wd = p.W;
    ra.nvec = ra.M / rp.vec;
    ra.cb = cb;
    ra.sb = sb;
    ra.K = rp.K;
    ra.items = p.C * rp.K;
    const char* st = knob(K_STAGGER);
    ra.stagger = st ? atoi(st) : 0;
    ra.prof = nullptr;
    ra.epoch = 0;
    ra.ctl_idle = kCtlIdle;
    ra.host_flag = resident_host_flag();
    ra.wait_ticks = resident_wait_ticks();  // (cnsn_set_wait_ms, else CNSN_WAIT_MS, else 5 s)
    const char* fi = knob(K_FAULT_INJECT);
    ra.fault = (fi && fi[0] == '1') ? 1 : 0;
    const char* xc = knob(K_XCD);
    ra.xcd = (xc && xc[0] == '1') ? 1 : 0;  // (measured slower: profiles/r05_xcd_clusters.md — off unless asked for)
    return ra;
}

// persistent grid: every workgroup resident, a whole number of clusters
template <typename Kern>
int grid_for(Kern kern, size_t lds, int K, int items) {
    // (asked once per kernel, LDS size and device: the query costs microseconds, and small sites are launch-bound)
    static std::mutex mu;
    static std::map<std::tuple<const void*, size_t, int>, int> known;  // (exact key: an over-estimate would over-size a persistent grid)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const auto key = std::make_tuple((const void*)kern, lds, dev);
    int occ = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = known.find(key);
        if (it != known.end()) {
            occ = it->second;
        } else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kBlock, lds) != hipSuccess) return 0;
            known[key] = occ;
        }
    }
    // MI355X admits min(API, 8, floor(800 / (ceil(sgpr/16)*16 + 16))) workgroups of 256 threads per CU
    // and the API over-reports by one when SGPRs are the limiter (MI355X_MICROARCH.md, "Residency").
    // Every resident kernel here uses 106-108 SGPRs (checked at build time) -> 6.
    if (occ > 6) occ = 6;
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] resident grid: occupancy %d/CU x %d CUs, K=%d, items=%d, lds=%zu\n", occ, cu_count(),
                K, items, lds);
    // Head-room (cnsn_set_headroom_cus(n) / CNSN_HEADROOM_CUS = n; the Python layer sets it under a process group): size the grid for n compute units fewer than the part has.  A persistent grid sized
    // for the whole chip next to a kernel that HOLDS compute units for longer than the launch (RCCL's channel kernels during a
    // large all-reduce) still completes — its clusters drain in order — but the workgroups that found no slot only start once
    // others have finished ALL their items: the launch takes up to twice as long.  With the grid sized for what is free, every
    // workgroup is resident from the start and the items are shared evenly (tests/test_gpu_foreign_kernel.py, 32 CUs held).
    int cus = cu_count();
    if (const int n = resident_headroom_cus()) cus = cus - n > 8 ? cus - n : 8;
    long g = (long)occ * cus;
    g = (g / K) * K;
    if (g > items) g = items;  // items is a multiple of K
    return (int)g;
}

// epi: the call carries the residual-block epilogue (EPI kernels; their AUTO rule is separate)
static inline ResPlan plan_impl(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, bool backward, bool epi,
                                bool post = false) {
    ResPlan rp{false, 0, 0, 0, 0};
    if (p.strategy == CNSN_STRATEGY_TWO_PASS || p.strategy == CNSN_STRATEGY_LOCAL || p.strategy == CNSN_STRATEGY_MONO ||
        has_chan_perm)
        return rp;
    if (resident_degraded()) return rp;  // a launch gave up and nobody re-armed since (cnsn_resident_rearm): not even when forced
    const int M = p.H * p.W;
    rp.vec = pick_vec(p.dtype, boxed ? p.W : M);
    if (!(rp.vec == 16 / elem_bytes(p.dtype) || (elem_bytes(p.dtype) == 2 && rp.vec == 4))) return rp;
    const int nvec = M / rp.vec;
    if (nvec <= 32) return rp;  // tiny planes: handled by the 16-lanes-per-plane streaming kernels
    const int need = (nvec + 63) / 64;
    for (const int nv : kBucketNv)
        if (nv >= need) {
            rp.nv = nv;
            rp.ppw = ppw_of(nv, backward, epi, elem_bytes(p.dtype));
            break;
        }
    if (rp.nv == 0) return rp;  // plane does not fit one wave's registers
    const int own = 4 * rp.ppw;
    rp.K = (p.N + own - 1) / own;
    if (res_lds_bytes(p.N, 6, own, BC_ROWS, true) > 64 * 1024) return rp;
    if (rp.K > 2 * cu_count()) return rp;
    // AUTO: use the resident kernels where they measured faster than the (non-temporal) two-pass kernels on
    // MI355X (profiles/r01_resident_tuning.md, last sweeps): every eligible fp32 shape, both directions; for
    // 16-bit tensors the plane classes listed below — the resident kernels are bound by the latency of the
    // cluster exchange, which halving the bytes does not shorten, while two-pass bf16 streams at 5.5 TB/s.
    // A register bucket more than 25 % larger than the plane needs is not worth it either.
    // CNSN_STRATEGY_RESIDENT forces the resident kernels wherever they are eligible.
    if (p.strategy == CNSN_STRATEGY_AUTO) {
        if (!resident_auto_enabled()) return rp;  // switched off, or a launch timed out earlier: degrade to two-pass
        if ((rp.nv - need) * 4 > need) return rp;
        // 16-bit (sweeps of round 1, profiles/r01_resident_tuning.md): up to 4 slots per lane (14x14 .. 44x44)
        // always; 7/8 slots (56x56, 64x64) un-boxed always; 13/16 slots only the un-boxed backward.
        // With crop boxes (round 4, tools/boxed_sweep.py -> profiles/r04_boxed_sweep.md, after the region select of the boxed
        // kernels became branch-free — until then two-pass won every 16-bit boxed call of these classes): 7 slots both
        // directions at every batch size (56x56 bf16 crop=both forward 0.254 vs 0.277 ms at N = 256, 0.038 vs 0.054 at N = 32);
        // 8 slots the backward (64x64: -10 % at N = 256, -24 % at N = 64) and the forward of CrossNorm alone (-5..-10 %); the
        // forward with SelfNorm was left to two-pass in round 4 (0.172 vs 0.180-0.192 ms at N = 256 in a loop of forwards only)
        const bool solo = !backward && !boxed && !p.cn_active && !(p.sn_active && p.sn_training);  // inference
        if (!epi && !solo && p.dtype != CNSN_F32) {
            const bool ok16 = rp.nv <= 4   ? true
                              : rp.nv == 7 ? true
                              : rp.nv == 8 ? true  // (round 5, tools/auto_audit.py: the forward with crop boxes AND SelfNorm too —
                                                   //  (16,2048,64,64) bf16 crop=both 0.354 -> 0.318 ms per call, (16,512,64,64)
                                                   //  0.118 -> 0.103: config 5's mode is single-touch in both directions now)
                                           : (backward && !boxed);
            if (!ok16) return rp;
        }
        if (epi && p.dtype != CNSN_F32 && rp.nv > 8) return rp;  // (those instantiations spill registers)
        // fp32 64x64 planes (16 slots) WITH crop boxes at small batches (segmentation: N = 16, K = 4 members per channel): two-pass
        // was 8-16 % faster while the boxed region select branched per element; since it is branch-free the cluster kernels win
        // (profiles/r04_boxed_sweep.md: (16,2048,64,64) forward 0.238 vs 0.279 ms, (16,512,64,64) 0.076 vs 0.081; the rule
        // that sent these calls to two-pass is gone.  The backward of CrossNorm alone at (16,512,64,64) measures 0.140 vs 0.115 in a
        // loop of backwards only — x and G of the previous call still in the 256 MB cache — and 0.178 vs 0.192 for the call in a
        // forward + backward loop, tools/auto_audit.py: the cluster kernels)
        // POST forward of the 16-bit 56x56 class: two-pass 0.283 vs 0.304 ms at (256,256,56,56) (profiles/r02_post_add.md)
        if (post && !backward && p.dtype != CNSN_F32 && rp.nv == 7 && p.sn_training) return rp;
    }
    rp.ok = true;
    return rp;
}

// post: the addend joins after the op (EPI only, un-boxed only — resident_fused_plan)
template <bool EPI>
int forward_impl(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* x,
                 const void* addend, int relu, const int64_t* perm, GateDev g, GateDev f, void* y, double* saved,
                 void* workspace, hipStream_t stream, bool post = false) {
    const ResPlan rp = plan_impl(p, boxed, false, false, EPI);
    if (!rp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    ResArgs ra = make_args(p, cb, sb, mid, rp);
    // nothing couples the planes of a channel (inference: SelfNorm on running statistics, no CrossNorm)
    const bool solo = !boxed && !p.cn_active && !(p.sn_active && p.sn_training);
    const int NG = boxed ? 6 : 2;
#ifdef CNSN_PROF  // tuning builds: time stamps land 4 MiB into the workspace (callers size it accordingly)
    if (knob(K_PROF)) ra.prof = (unsigned long long*)((char*)workspace + (4u << 20));
#endif
    const size_t lds = res_lds_bytes(p.N, NG, 4 * rp.ppw, FC_ROWS, false);
    std::optional<ResidentChain> chain;  // cluster grids of different streams never overlap
    if (!solo) chain.emplace(stream);  // (the exchange area is taken inside the chain: a context's wrap-around clear is ordered like a launch)
    const ExchangeArea ea = solo ? ExchangeArea{workspace, 0u}
                                 : resident_exchange_area(p, kCtlBytes + (size_t)p.N * p.C * NG * 8, workspace, stream);
    ra.epoch = ea.epoch;
    ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
    const size_t fill_bytes = kCtlBytes + (size_t)p.N * p.C * (NG / 2) * 8;  // control block + granules (workspace form)
    PongArea pong{nullptr, nullptr, 0u, false};  // (a granule region of the context instead of workspace + fill: resident_pong_acquire)
    const bool use_pong = !solo && !ea.epoch && resident_pong_acquire(p, fill_bytes, stream, &pong);
    void* area = use_pong ? pong.base : ea.base;
    unsigned* ctl = (unsigned*)area;
    unsigned long long* gran = (unsigned long long*)((char*)area + kCtlBytes);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_res<false, EPI>(p.dtype, rp.vec, rp.nv, [&](auto tt, auto vt, auto nt, auto pt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value, PPW = decltype(pt)::value;
        auto launch = [&](auto kern) {
            const int grid = grid_for(kern, lds, rp.K, ra.items);
            if (grid < rp.K) return;
            hipError_t e = hipSuccess;
            if (solo) {  // no cluster, nothing to wait for
                kern<<<grid, kBlock, lds, stream>>>(ra, (const T*)x, (T*)y, perm, g, f, gran, saved, ctl, (const T*)addend,
                                                    relu, nullptr, 0u, *pin);
            } else {
                if (!ea.epoch && (!use_pong || pong.need_fill)) e = hipMemsetAsync(area, 0xff, fill_bytes, stream);  // 'empty' granules, idle control word
                if (e != hipSuccess) {
                    status = (int)e;
                    return;
                }
                kern<<<grid, kBlock, lds, stream>>>(ra, (const T*)x, (T*)y, perm, g, f, gran, saved, ctl, (const T*)addend,
                                                    relu, pong.clear, pong.clear_qwords, *pin);
            }
            e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
            if (use_pong && e == hipSuccess) resident_pong_commit(p, fill_bytes);
        };
        if constexpr (EPI) {
            if (post) {  // (never boxed)
                if (solo)
                    launch(resident_fwd_kernel<T, VEC, NV, PPW, false, true, true, true>);
                else
                    launch(resident_fwd_kernel<T, VEC, NV, PPW, false, true, false, true>);
                return;
            }
        }
        if (boxed)
            launch(resident_fwd_kernel<T, VEC, NV, PPW, true, EPI>);
        else if (solo)
            launch(resident_fwd_kernel<T, VEC, NV, PPW, false, EPI, true>);
        else
            launch(resident_fwd_kernel<T, VEC, NV, PPW, false, EPI>);
    });
    return status;
}

template <bool EPI>
int backward_impl(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy,
                  const void* x, const void* addend, int relu, const int64_t* perm, GateDev g, GateDev f,
                  const double* saved, void* dx, GateGradDev dg, GateGradDev df, void* workspace, hipStream_t stream,
                  bool post = false, void* d_addend = nullptr) {
    const ResPlan rp = plan_impl(p, boxed, false, true, EPI);
    if (!rp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    ResArgs ra = make_args(p, cb, sb, mid, rp);
    const int NS = boxed ? 4 : 2;
#ifdef CNSN_PROF
    if (knob(K_PROF)) ra.prof = (unsigned long long*)((char*)workspace + (4u << 20));
#endif
    const size_t lds = res_lds_bytes(p.N, NS, 4 * rp.ppw, BC_ROWS, true);
    ResidentChain chain(stream);  // cluster grids of different streams never overlap
    // (the exchange area is taken inside the chain: a context's wrap-around clear is ordered like a launch)
    const ExchangeArea ea = resident_exchange_area(p, kCtlBytes + (size_t)p.N * p.C * NS * 8, workspace, stream);
    ra.epoch = ea.epoch;
    ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
    const size_t fill_bytes = kCtlBytes + (size_t)p.N * p.C * (NS / 2) * 8;
    PongArea pong{nullptr, nullptr, 0u, false};  // (a granule region of the context instead of workspace + fill: resident_pong_acquire)
    const bool use_pong = !ea.epoch && resident_pong_acquire(p, fill_bytes, stream, &pong);
    void* area = use_pong ? pong.base : ea.base;
    unsigned* ctl = (unsigned*)area;
    unsigned long long* gran = (unsigned long long*)((char*)area + kCtlBytes);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_res<true, EPI>(p.dtype, rp.vec, rp.nv, [&](auto tt, auto vt, auto nt, auto pt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value, PPW = decltype(pt)::value;
        auto launch = [&](auto kern) {
            const int grid = grid_for(kern, lds, rp.K, ra.items);
            if (grid < rp.K) return;
            hipError_t e = hipSuccess;
            if (!ea.epoch && (!use_pong || pong.need_fill)) e = hipMemsetAsync(area, 0xff, fill_bytes, stream);  // 'empty' granules
            if (e != hipSuccess) {
                status = (int)e;
                return;
            }
            kern<<<grid, kBlock, lds, stream>>>(ra, (const T*)gy, (const T*)x, (T*)dx, perm, g, f, dg, df, gran,
                                                saved, ctl, (const T*)addend, relu, (T*)d_addend, pong.clear, pong.clear_qwords, *pin);
            e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
            if (use_pong && e == hipSuccess) resident_pong_commit(p, fill_bytes);
        };
        if constexpr (EPI) {
            if (post) {
                launch(resident_bwd_kernel<T, VEC, NV, PPW, false, true, true>);
                return;
            }
        }
        if (boxed)
            launch(resident_bwd_kernel<T, VEC, NV, PPW, true, EPI>);
        else
            launch(resident_bwd_kernel<T, VEC, NV, PPW, false, EPI>);
    });
    return status;
}

}  // namespace reshost
}  // namespace cnsn
