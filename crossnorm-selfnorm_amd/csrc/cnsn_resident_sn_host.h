// Host side shared by the two translation units of the SelfNorm-only cluster kernels: eligibility, geometry.
#pragma once
#include "cnsn_fused_stream_kernels.h"
#include "cnsn_env.h"
#include "cnsn_resident_host.h"
#include "cnsn_resident_sn.h"
#include "cnsn_resident_sn_kernels.h"

namespace cnsn {
namespace snxhost {

constexpr size_t kLdsPerCu = 160 * 1024;

// planes per wave: the item in flight (x [+ addend]; G, x [+ addend]) has to fit the registers of 3-4 (forward) / 2-3
// (backward) workgroups per CU next to the kept part of the parked item
#ifndef SNX_NV2_PPW
#define SNX_NV2_PPW 4       // two-slot planes (28x28 in 16 bits) with the epilogue
#endif
#ifndef SNX_NV2_PPW_PLAIN
#define SNX_NV2_PPW_PLAIN 8 // ... without it (measured after the algebra went lane-parallel: 8 planes per wave -6 % / -2 %
#endif                      //     against 4, 2 planes per wave +14 % / +19 %: the fewer members the better)
#ifndef SNX_NV7_PPW16
#define SNX_NV7_PPW16 2     // 16-bit 56x56 class, forward without epilogue: planes per wave
#endif
#ifndef SNX_NV1_PPW8
#define SNX_NV1_PPW8 16     // one-slot planes of 8-byte vectors (16-bit 14x14): planes per wave
#endif
#ifndef SNX_NV1_PPW
#define SNX_NV1_PPW 8       // one-slot planes of 16-byte vectors (fp32 14x14)
#endif
#ifndef SNX_NV4_PPW_EPI
#define SNX_NV4_PPW_EPI 2   // four-slot planes (fp32 28x28, 40x40 in 16 bits), forward with the epilogue
#endif
#ifndef SNX_NV4_BWD_PPW
#define SNX_NV4_BWD_PPW 4   // ... backward without the epilogue ((128,256,40,40) bf16: 0.091 -> 0.071 ms; fp32 28x28 level); with it: 2
#endif
#ifndef SNX_NV7_PPW_EPI
#define SNX_NV7_PPW_EPI 2   // seven-slot planes in 16 bits (56x56), forward with the epilogue: 2 planes per wave -6 % at N = 256
#endif                      // (0.266 -> 0.249 ms), +3 % at N = 96; fp32 (40x40): 1
#ifndef SNX_NV7_BWD_PPW
#define SNX_NV7_BWD_PPW 2   // ... backward without the epilogue in 16 bits (-2 %); with it (does not fit) and fp32: 1
#endif
#ifndef SNX_NV13_FWD_PPW
#define SNX_NV13_FWD_PPW 1  // thirteen-slot planes (fp32 56x56), forward without the epilogue
#endif
constexpr int fwd_ppw(int nv, bool epi, int elem_bytes, int vb = 16) {
    return nv == 1   ? (vb == 8 ? SNX_NV1_PPW8 : SNX_NV1_PPW)
           : nv == 2 ? (epi ? SNX_NV2_PPW : SNX_NV2_PPW_PLAIN)
           : nv == 4 ? (epi ? SNX_NV4_PPW_EPI : 4)
           : nv == 7 ? (elem_bytes == 2 ? (epi ? SNX_NV7_PPW_EPI : SNX_NV7_PPW16) : 1)
           : nv == 13 ? (epi ? 1 : SNX_NV13_FWD_PPW)
                     : 1;
}
constexpr int bwd_ppw(int nv, bool epi, int elem_bytes, int vb = 16) {  // (one-slot planes with the epilogue: 3 x 16 planes in flight do not fit)
    return nv == 1 ? (vb == 8 ? (epi ? SNX_NV1_PPW8 / 2 : SNX_NV1_PPW8) : SNX_NV1_PPW)
           : nv == 2 ? (epi ? SNX_NV2_PPW : SNX_NV2_PPW_PLAIN)
           : nv == 4 ? (epi ? 2 : SNX_NV4_BWD_PPW)
           : nv == 7 ? ((!epi && elem_bytes == 2) ? SNX_NV7_BWD_PPW : 1)
                     : 1;
}

// classes of the CrossNorm-capable backward that are compiled: 16-byte vectors, planes of 7 and more register slots (40x40 fp32,
// 56x56 in 16 bits and larger).  The smaller classes hold 4-16 planes per wave: their CrossNorm records (own rows + the
// borrower's, 128 bytes per plane, two items) do not fit next to the parked item at three workgroups per CU, and one-slot
// planes have the channel-in-registers kernels, which take CrossNorm themselves (cnsn_mono_cn_kernels.h)
constexpr bool snx_cn_built(int nv, int eb, int vb) { return vb == 16 && nv >= 7 && (eb == 4 || eb == 2); }
// AUTO rule of the CrossNorm-capable backward (plan_impl, cn): register bucket, element bytes, batch.  Same-box A/B against the
// general (pipelined) cluster backward, HIP-event intervals of the backward call (profiles/r04_cn_partial_moments.md):
//   fp32  40x40 (7 slots) -4 %, 56x56 (13) level at N = 256 on a box whose memory floor the general kernel already reaches,
//         -3 % at N = 96, 64x64 (16) -6..-12 %                                   -> every fp32 class
//   16 bit 56x56 (7 slots) -6 % at N = 256, -5 % at N = 96; 64x64 (8 slots) -9 % at N = 16 but +3..+6 % at N = 64
//         (and level or +4..+8 % for the whole call at N = 16 in tools/auto_audit.py on another box)
//         -> 7 slots always; 8 / 13 / 16 slots: the general kernels
constexpr bool snx_cn_auto(int nv, int eb, int N) { return eb == 4 || nv == 7; }
// ... with crop boxes (profiles/r04_cn_partial_moments.md, "crop boxes"): fp32 56x56 -3 % at N = 256 (against the pipelined boxed
// kernel), -7 % at N = 96, 40x40 level, 64x64 at N = 16 -26 % (against two-pass); 16-bit 56x56 -3 % at N = 256 against TWO-PASS
// (3 instead of 5 tensor passes, but VALU-bound: 12 vector instructions per element in the apply), +3 % at N = 96.
// Since the region select is branch-free (profiles/r04_boxed_sweep.md): 16-bit 56x56 0.302 vs 0.354 ms at N = 256, 0.158-0.164
// vs 0.169 (general cluster kernel) / 0.183 (two-pass) at N = 128, 0.126 vs 0.134 / 0.140 at N = 96 -> every batch size; 64x64
// (8 slots) 0.217 vs 0.209 at N = 256, 0.121 vs 0.094 at N = 64 -> the general kernels
constexpr bool snx_cn_boxed_auto(int nv, int eb, int N) { return eb == 4 || nv == 7; }

// CNSN_SNX=0: never; 2: wherever instantiated (tests); default 1: AUTO rule
inline int snx_mode() {
    const char* e = knob(K_SNX);
    return e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1;
}

template <bool BWD, bool EPI, typename F>
bool dispatch_snx(int dtype, int vec, int nv, F&& f) {
    auto by_nv = [&](auto tt, auto vt) -> bool {
        using T = typename decltype(tt)::type;
        constexpr int EB = (int)sizeof(T);
        switch (nv) {
            case 2: f(tt, vt, IntTag<2>{}, IntTag<(BWD ? bwd_ppw(2, EPI, EB) : fwd_ppw(2, EPI, EB))>{}); return true;
            case 4: f(tt, vt, IntTag<4>{}, IntTag<(BWD ? bwd_ppw(4, EPI, EB) : fwd_ppw(4, EPI, EB))>{}); return true;
            case 7: f(tt, vt, IntTag<7>{}, IntTag<(BWD ? bwd_ppw(7, EPI, EB) : fwd_ppw(7, EPI, EB))>{}); return true;
            case 8: f(tt, vt, IntTag<8>{}, IntTag<(BWD ? bwd_ppw(8, EPI, EB) : fwd_ppw(8, EPI, EB))>{}); return true;
            case 13: f(tt, vt, IntTag<13>{}, IntTag<(BWD ? bwd_ppw(13, EPI, EB) : fwd_ppw(13, EPI, EB))>{}); return true;
            case 16:
                if constexpr (BWD && EPI) return false;  // 48 slots in flight: does not fit
                else { f(tt, vt, IntTag<16>{}, IntTag<1>{}); return true; }
            default: return false;
        }
    };
    // one-slot planes (33..64 vectors, the 14x14 class): many planes per wave
    auto one_slot = [&](auto tt, auto vt) -> bool {
        using T = typename decltype(tt)::type;
        constexpr int EB = (int)sizeof(T), VB = decltype(vt)::value * EB;
        f(tt, vt, IntTag<1>{}, IntTag<(BWD ? bwd_ppw(1, EPI, EB, VB) : fwd_ppw(1, EPI, EB, VB))>{});
        return true;
    };
    if (nv == 1) {
        if (dtype == CNSN_F32 && vec == 4) return one_slot(TypeTag<float>{}, IntTag<4>{});
        if (dtype == CNSN_BF16 && vec == 4) return one_slot(TypeTag<bf16_t>{}, IntTag<4>{});
        if (dtype == CNSN_F16 && vec == 4) return one_slot(TypeTag<_Float16>{}, IntTag<4>{});
        return false;
    }
    if (dtype == CNSN_F32 && vec == 4) return by_nv(TypeTag<float>{}, IntTag<4>{});
    if (dtype == CNSN_BF16 && vec == 8) return by_nv(TypeTag<bf16_t>{}, IntTag<8>{});
    if (dtype == CNSN_F16 && vec == 8) return by_nv(TypeTag<_Float16>{}, IntTag<8>{});
    return false;
}

inline size_t lds_bytes(int K, int own, int np, bool backward, int vb, bool cn = false, int N = 0) {
    return backward ? snx_bwd_lds_bytes(K, own, np, vb, cn, N) : snx_fwd_lds_bytes(K, own, np, vb);
}

// CNSN_SNXCN=0: the CrossNorm-capable backward never; 2: wherever instantiated (tests); default: AUTO rule
inline int snx_cn_mode() {
    const char* e = knob(K_SNXCN);
    return e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1;
}

// cn: the CrossNorm-capable backward (un-boxed CrossNorm in front of SelfNorm, no epilogue: round 4)
inline SnxPlan plan_impl(const cnsn_problem_t& p, bool boxed, int add, int relu, bool backward, bool cn = false) {
    SnxPlan none{false, 0, 0, 0, 0, 0};
    const int mode = cn ? (snx_mode() == 0 ? 0 : snx_cn_mode()) : snx_mode();
    if (mode == 0) return none;
    if (!(p.strategy == CNSN_STRATEGY_AUTO || p.strategy == CNSN_STRATEGY_RESIDENT)) return none;
    if ((boxed && !cn) || (p.cn_active != 0) != cn || !p.sn_active || !p.sn_training || p.sn_two) return none;
    if (cn && (!backward || add != ADD_NONE || relu || p.N > kPermInlineMax)) return none;
    if (cn && (pick_vec(p.dtype, boxed ? p.W : p.H * p.W) * elem_bytes(p.dtype) != 16 ||
               (p.H * p.W / pick_vec(p.dtype, boxed ? p.W : p.H * p.W) + 63) / 64 < 5))
        return none;  // (snx_cn_built: buckets of 7 slots and more)
    if (!(add == ADD_NONE || add == ADD_PRE)) return none;
    if (resident_degraded()) return none;
    if (p.strategy == CNSN_STRATEGY_AUTO && !resident_auto_enabled()) return none;
    if ((long)p.N * p.C < 16) return none;  // (the exchange area is sized against the two-pass workspace: see cnsn_resident_sn.hip)
    const bool epi = add != ADD_NONE || relu;
    const int M = p.H * p.W, eb = elem_bytes(p.dtype);
    SnxPlan sp = none;
    sp.vec = pick_vec(p.dtype, boxed ? p.W : M);  // (crop boxes: a vector must not straddle two rows)
    if ((size_t)p.N * p.C * M * eb >= ((size_t)1 << 30)) return none;  // one descriptor per tensor, 32-bit offsets that must not wrap: see PlaneIo
    const int vb = sp.vec * eb;
    const int nvec = M / sp.vec;
    // 16-byte vectors; one-slot planes (the 14x14 class) also with 8-byte ones.  Planes of at most 32 vectors stay with
    // the channel-in-registers kernels
    if (!(vb == 16 || (vb == 8 && nvec <= 64 && eb == 2)) || nvec <= 32) return none;
    const int need = (nvec + 63) / 64;
    for (const int nv : {1, 2, 4, 7, 8, 13, 16})
        if (nv >= need) {
            sp.nv = nv;
            break;
        }
    if (sp.nv == 0) return none;
    if (backward && epi && sp.nv == 16) return none;
    if (mode != 2 && (sp.nv - need) * 4 > need) return none;  // a register bucket far larger than the plane
    sp.ppw = backward ? bwd_ppw(sp.nv, epi, eb, vb) : fwd_ppw(sp.nv, epi, eb, vb);
    const int own = 4 * sp.ppw;
    sp.K = (p.N + own - 1) / own;
    if (sp.K > 2 * reshost::cu_count() || sp.K > 1024) return none;
    const int slots = (backward ? 2 : 1) * sp.ppw * sp.nv;
    const int wg_per_cu = backward ? snx_bwd_waves(slots, epi, vb) : snx_fwd_waves(slots, epi, eb, vb);
    const long grid_max = ((long)wg_per_cu * reshost::cu_count() / sp.K) * sp.K;
    if (grid_max < sp.K) return none;
    const size_t budget = (kLdsPerCu / wg_per_cu) & ~(size_t)511;
    const int first_keep = slots - (backward ? snx_bwd_keep(slots, epi, vb, cn, cn && boxed) : snx_fwd_keep(slots, vb, epi && eb == 2));
    int np = slots;
    if (!backward && sp.ppw == 1 && nvec % 64 != 0 && nvec % 64 <= 32) np = slots - 1;  // a last, partly filled slot stays in registers
    if (np < first_keep) np = first_keep;
    while (np >= first_keep && lds_bytes(sp.K, own, np, backward, vb, cn, p.N) > budget) --np;
    if (np < first_keep) return none;
    // AUTO / forced-resident without CNSN_SNX=2: where these kernels measured faster than the general resident kernels
    // on MI355X (profiles/r03_sn_cluster.md, same-process A/B): every call WITH the residual-block epilogue (the general
    // kernels are not pipelined there: 56x56 bf16 block forward -18 %, backward -32 %); without it every class except
    // the forward of the fp32 28x28 class (+3 % against the general pipelined forward) and the backward of the 16-bit
    // 56x56 class at small batches (N = 96: +9 %, N = 256: -7 %).  One-slot planes: see resident_sn_prefers.
    if (mode != 2 && !epi) {
        if (!backward && eb == 4 && sp.nv == 4) return none;
        // (round 3 kept the 16-bit 56x56 backward at N < 128 with the general kernels: +9 % at N = 96 then; since the launches
        //  run back to back — round 4 — it is 7 % FASTER for the whole call on both audited boxes: the exclusion is gone)
    }
    // the CrossNorm-capable backward, AUTO: the classes measured against the general (pipelined) cluster backward on MI355X
    // (profiles/r04_cn_partial_moments.md)
    if (cn && mode != 2 && !(boxed ? snx_cn_boxed_auto(sp.nv, eb, p.N) : snx_cn_auto(sp.nv, eb, p.N))) return none;
    sp.npark = np;
    sp.ok = true;
    return sp;
}

inline ResArgs make_args(const cnsn_problem_t& p, const MidArgs& mid, const SnxPlan& sp) {
    ResPlan rp{true, sp.vec, sp.nv, sp.ppw, sp.K};
    const Box whole{0, 0, p.H, p.W};
    return reshost::make_args(p, whole, whole, mid, rp);
}

// exchange areas, in granules of 8 bytes: round A per member, round B per wave (backward only), the per-plane sums of the
// CrossNorm-capable backward (two tagged granules per plane)
inline size_t tagged_bytes(const cnsn_problem_t& p, int K, bool backward, bool cn = false, bool boxed = false) {
    return kCtlBytes + (size_t)p.C * K * (backward ? 4 + 8 : 4) * 8 + 1024 + (cn ? (size_t)p.N * p.C * (boxed ? 4 : 2) * 8 + 256 : 0);
}

}  // namespace snxhost
}  // namespace cnsn
