// Host side of the pipelined cluster-resident forward (cnsn_resident_pipe_kernels.h): eligibility, how many slots of
// an item are parked in LDS, launch.  Offered for the op alone (no epilogue), training or CrossNorm-coupled calls, planes
// of 7+ register slots (40x40 fp32 / 56x56 16-bit and larger) — the classes whose items are large enough (>= 25 KB per
// workgroup) for the exposed cluster wait to be the bottleneck.
#include "cnsn_resident_host.h"

#include <cstring>
#include "cnsn_env.h"
#include "cnsn_resident_pipe_kernels.h"

namespace cnsn {

namespace {

constexpr size_t kLdsPerCu = 160 * 1024;

// f(TypeTag<T>, IntTag<VEC>, IntTag<NV>, IntTag<PPW>)
template <typename F>
bool dispatch_pipe(int dtype, int vec, int nv, F&& f) {
    auto by_nv = [&](auto tt, auto vt) -> bool {
        using T = typename decltype(tt)::type;
        switch (nv) {
            case 2: f(tt, vt, IntTag<2>{}, IntTag<reshost::ppw_of(2, false, false, (int)sizeof(T))>{}); return true;
            case 4: f(tt, vt, IntTag<4>{}, IntTag<reshost::ppw_of(4, false, false, (int)sizeof(T))>{}); return true;
            case 7: f(tt, vt, IntTag<7>{}, IntTag<reshost::ppw_of(7, false, false, (int)sizeof(T))>{}); return true;
            case 8: f(tt, vt, IntTag<8>{}, IntTag<reshost::ppw_of(8, false, false, (int)sizeof(T))>{}); return true;
            case 13: f(tt, vt, IntTag<13>{}, IntTag<1>{}); return true;
            case 16: f(tt, vt, IntTag<16>{}, IntTag<1>{}); return true;
            default: return false;
        }
    };
    if (dtype == CNSN_F32 && vec == 4) return by_nv(TypeTag<float>{}, IntTag<4>{});
    if (dtype == CNSN_BF16 && vec == 8) return by_nv(TypeTag<bf16_t>{}, IntTag<8>{});
    if (dtype == CNSN_F16 && vec == 8) return by_nv(TypeTag<_Float16>{}, IntTag<8>{});
    return false;
}

// Which (element size, slots, crop boxes) classes of the pipelined forward are BUILT: the ones an AUTO rule of
// resident_pipe_plan can pick (round 6, the variant budget: 15 of the 36 instantiations were reachable with CNSN_PIPE=2
// only — fp32 un-boxed 8 / 16 slots, fp32 boxed 2 / 7 / 8, 16-bit un-boxed 16, 16-bit boxed 4 / 8 / 13 / 16 — every one of
// them slower than the plain cluster kernel where it was measured, most of them spilling; those calls run the plain kernel).
constexpr bool pipe_class_built(int elem, int nv, bool boxed) {
    if (elem == 4) return boxed ? (nv == 4 || nv == 13 || nv == 16) : (nv == 2 || nv == 4 || nv == 7 || nv == 13);
    return boxed ? (nv == 2 || nv == 7) : (nv == 2 || nv == 4 || nv == 7 || nv == 8 || nv == 13);
}

// CNSN_PIPE=0: never; CNSN_PIPE=2: also for grids with fewer than three items per workgroup (tests)
int pipe_mode() {
    const char* e = knob(K_PIPE);
    return e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1;
}

}  // namespace

// rp.ok: the pipelined forward takes the call; *npark = slots of an item that go to LDS
ResPlan resident_pipe_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int* npark) {
    ResPlan none{false, 0, 0, 0, 0};
    const int mode = pipe_mode();
    if (mode == 0) return none;
    ResPlan rp = reshost::plan_impl(p, boxed, has_chan_perm, false, false);
    if (!rp.ok) return none;
    if (!boxed && !p.cn_active && !(p.sn_active && p.sn_training)) return none;  // inference: nothing to wait for
    const int vb = rp.vec * elem_bytes(p.dtype);
    if (vb != 16 || rp.nv < 2) return none;
    if (!pipe_class_built(elem_bytes(p.dtype), rp.nv, boxed)) return none;  // (not even when forced)
    if ((size_t)p.N * p.C * p.H * p.W * elem_bytes(p.dtype) >= ((size_t)1 << 30)) return none;  // one descriptor per tensor, 32-bit offsets that must not wrap: see PlaneIo
    const int slots = rp.ppw * rp.nv, nvec = p.H * p.W / rp.vec;
    const int wg_per_cu = pipe_fwd_waves(slots);
    const int grid_max = (wg_per_cu * reshost::cu_count() / rp.K) * rp.K;
    // fewer than three items per workgroup: no pipeline to fill — except 16-bit 2-slot planes with crop boxes, where this kernel's
    // boxed code is simply the better one: (128,32,32,32) bf16 crop=both 0.077 against 0.086 / 0.104 / 0.077 ms with the plain
    // cluster kernel on three audits (profiles/r04_auto_audit_after.md, r05_auto_audit_box1.md, r05_auto_audit_box2_after.md)
    const bool few_items = (long)p.C * rp.K < 3l * grid_max;
    const bool small_boxed16 = boxed && elem_bytes(p.dtype) == 2 && rp.nv == 2;
    if (grid_max < rp.K || (mode != 2 && few_items && !small_boxed16)) return none;
    // slots worth parking: the full ones of every plane held (a last, partly filled slot may as well stay in registers)
    const int keep_max = slots < kPipeKeep ? slots : kPipeKeep;
    const size_t budget = (kLdsPerCu / wg_per_cu) & ~(size_t)511;
    int np = slots;
    if (rp.ppw == 1 && nvec % 64 != 0 && nvec % 64 <= 32) np = slots - 1;
    while (np >= slots - keep_max && pipe_lds_bytes(p.N, boxed ? 6 : 2, 4 * rp.ppw, np, vb, p.cn_active != 0) > budget) --np;
    if (np < slots - keep_max) return none;
    // AUTO / forced-resident without CNSN_PIPE=2: where it measured faster than the plain kernel on MI355X
    // (profiles/r02_pipelined_resident.md): un-boxed calls of the 2-, 4-, 7- and 13-slot classes (28x28: 2-8 %, 40x40
    // 16-bit: 20 %, 56x56: 9 %).  With crop boxes the gather is three times as long and the kernel is VALU-bound (0.53 vs
    // 0.46 ms at the north-star shape); the 8-slot class measured 4 % slower, the 16-slot class spills.
    if (mode != 2) {
        // With crop boxes (round 4, profiles/r04_boxed_sweep.md, after the branch-free region select): fp32 56x56 at every batch
        // size (0.375 vs 0.402 ms at N = 256, 0.179 vs 0.189 at N = 128, 0.047 vs 0.053 at N = 32), fp32 64x64 from N = 64
        // (0.234 vs 0.258 at N = 256, 0.123 vs 0.132 at N = 64; at N = 16 the plain kernel: 0.238 vs 0.304), 16-bit 56x56 at
        // every batch size (0.254 vs 0.294, 0.128 vs 0.143, 0.038 vs 0.048); 16-bit 64x64 the plain kernel (0.158 vs 0.164).
        if (boxed) {
            const bool f32 = elem_bytes(p.dtype) == 4;
            // (fp32 28x28 — 4 slots — too since round 5: -3.4 / -4.1 % per call at (256,512,28,28) and (96,512,28,28) on both audited boxes)
            if (!((f32 && rp.nv == 13) || (f32 && rp.nv == 16 && p.N >= 64) || (f32 && rp.nv == 4) || (!f32 && rp.nv == 7) ||
                  (small_boxed16 && few_items)))
                return none;
        } else if (!(rp.nv == 2 || rp.nv == 4 || rp.nv == 7 || rp.nv == 13 || (rp.nv == 8 && elem_bytes(p.dtype) == 2)))
            return none;  // (8 slots in 16 bits: round 5, after the forward's register pass — (16,512,64,64) bf16 0.105 -> 0.093 ms per
                          //  call, (16,2048,64,64) 0.321 -> 0.300: tools/auto_audit.py)
    }
    if (npark) *npark = np;
    return rp;
}

int resident_pipe_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* x,
                          const int64_t* perm, GateDev g, GateDev f, void* y, double* saved, void* workspace,
                          hipStream_t stream) {
    int npark = 0;
    const ResPlan rp = resident_pipe_plan(p, boxed, false, &npark);
    if (!rp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    ResArgs ra = reshost::make_args(p, cb, sb, mid, rp);
    const int NG = boxed ? 6 : 2;
#ifdef CNSN_PROF
    if (knob(K_PROF)) ra.prof = (unsigned long long*)((char*)workspace + (4u << 20));
#endif
    const size_t lds = pipe_lds_bytes(p.N, NG, 4 * rp.ppw, npark, rp.vec * elem_bytes(p.dtype), p.cn_active != 0);
    ResidentChain chain(stream);  // cluster grids of different streams never overlap
    // (the exchange area is taken inside the chain: a context's wrap-around clear is ordered like a launch)
    const ExchangeArea ea = resident_exchange_area(p, kCtlBytes + (size_t)p.N * p.C * NG * 8 + 256, workspace, stream);
    ra.epoch = ea.epoch;
    ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
    const size_t fill_bytes = kCtlBytes + (size_t)p.N * p.C * (NG / 2) * 8;
    // workspace form (large tensors): through a granule region of the context that the previous launch left clean, if there
    // is one — no fill launch (resident_pong_acquire)
    PongArea pong{nullptr, nullptr, 0u, false};
    const bool use_pong = !ea.epoch && resident_pong_acquire(p, fill_bytes, stream, &pong);
    void* area = use_pong ? pong.base : ea.base;
    unsigned* ctl = (unsigned*)area;
    unsigned long long* gran = (unsigned long long*)((char*)area + kCtlBytes);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_pipe(p.dtype, rp.vec, rp.nv, [&](auto tt, auto vt, auto nt, auto pt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value, PPW = decltype(pt)::value;
        auto launch = [&](auto kern) {
            if (!allow_dynamic_lds(kern, lds)) return;
            const int grid = reshost::grid_for(kern, lds, rp.K, ra.items);
            if (grid < rp.K) return;
            hipError_t e = hipSuccess;
            if (!ea.epoch && (!use_pong || pong.need_fill)) e = hipMemsetAsync(area, 0xff, fill_bytes, stream);
            if (e != hipSuccess) {
                status = (int)e;
                return;
            }
            PipeFwdKargs<T> ka{ra, npark, (const T*)x, (T*)y, perm, g, f, gran, saved, ctl, pong.clear, pong.clear_qwords, {}};
            ka.pin.on = pin->on;  // (the index table only when it is in use: 2 KB of kernel arguments)
            if (pin->on) memcpy(ka.pin.v, pin->v, sizeof(unsigned short) * (size_t)p.N);
            kern<<<grid, kBlock, lds, stream>>>(ka);
            e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
            if (use_pong && e == hipSuccess) resident_pong_commit(p, fill_bytes);
        };
        if (boxed) {
            if constexpr (pipe_class_built((int)sizeof(T), NV, true)) launch(resident_fwd_pipe_kernel<T, VEC, NV, PPW, true>);
        } else {
            if constexpr (pipe_class_built((int)sizeof(T), NV, false)) launch(resident_fwd_pipe_kernel<T, VEC, NV, PPW, false>);
        }
    });
    return status;
}


// ---- backward ----------------------------------------------------------------------------------------------------
namespace {

// planes per wave of the pipelined backward (its own choice: G and x of an item are both on chip twice over)
constexpr int pipe_bwd_ppw(int nv) { return nv == 2 ? 4 : (nv == 4 ? 2 : 1); }

// f(TypeTag<T>, IntTag<VEC>, IntTag<NV>, IntTag<PPW>)
template <typename F>
bool dispatch_pipe_bwd(int dtype, int vec, int nv, F&& f) {
    auto by_nv = [&](auto tt, auto vt) -> bool {
        // Only the 13-slot class is built.  The kernel is generic in (NV, PPW); the 16-slot items of the smaller
        // classes — (2,4), (4,2), (7,1), (8,1) — were tried at three workgroups per CU: 50-60 spilled VGPRs at the
        // 168-register budget and 5-30 % slower than the plain kernels; the 7-slot class with two planes per wave (28 slots, two workgroups per CU): +5 % in 16
        // bits, -1 % in fp32 (profiles/r02_pipelined_resident.md).
        switch (nv) {
            case 13: f(tt, vt, IntTag<13>{}, IntTag<pipe_bwd_ppw(13)>{}); return true;
            default: return false;
        }
    };
    if (dtype == CNSN_F32 && vec == 4) return by_nv(TypeTag<float>{}, IntTag<4>{});
    if (dtype == CNSN_BF16 && vec == 8) return by_nv(TypeTag<bf16_t>{}, IntTag<8>{});
    if (dtype == CNSN_F16 && vec == 8) return by_nv(TypeTag<_Float16>{}, IntTag<8>{});
    return false;
}
}  // namespace

ResPlan resident_pipe_bwd_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int* npark) {
    ResPlan none{false, 0, 0, 0, 0};
    const int mode = pipe_mode();
    if (mode == 0) return none;
    ResPlan rp = reshost::plan_impl(p, boxed, has_chan_perm, true, false);
    if (!rp.ok) return none;
    const int vb = rp.vec * elem_bytes(p.dtype);
    if (vb != 16 || rp.nv != 13) return none;
    rp.ppw = pipe_bwd_ppw(rp.nv);
    rp.K = (p.N + 4 * rp.ppw - 1) / (4 * rp.ppw);
    if (rp.K > 2 * reshost::cu_count()) return none;
    const int slots = 2 * rp.ppw * rp.nv;
    const int wg_per_cu = pipe_bwd_waves(slots);
    const int grid_max = (wg_per_cu * reshost::cu_count() / rp.K) * rp.K;
    if (grid_max < rp.K || (mode != 2 && (long)p.C * rp.K < 3l * grid_max)) return none;
    const size_t budget = (kLdsPerCu / wg_per_cu) & ~(size_t)511;
    int np = slots;
    while (np >= pipe_bwd_first_keep(slots) && pipe_bwd_lds_bytes(p.N, boxed ? 4 : 2, 4 * rp.ppw, np, vb) > budget) --np;
    if (np < pipe_bwd_first_keep(slots)) return none;
    if (mode != 2) {
        if (boxed && !(elem_bytes(p.dtype) == 4 && p.N >= 192)) return none;  // (as the forward)
    }
    if (npark) *npark = np;
    return rp;
}

int resident_pipe_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, const void* gy,
                           const void* x, const int64_t* perm, GateDev g, GateDev f, const double* saved, void* dx,
                           GateGradDev dg, GateGradDev df, void* workspace, hipStream_t stream) {
    int npark = 0;
    const ResPlan rp = resident_pipe_bwd_plan(p, boxed, false, &npark);
    if (!rp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    ResArgs ra = reshost::make_args(p, cb, sb, mid, rp);
    const int NS = boxed ? 4 : 2;
#ifdef CNSN_PROF
    if (knob(K_PROF)) ra.prof = (unsigned long long*)((char*)workspace + (4u << 20));
#endif
    const size_t lds = pipe_bwd_lds_bytes(p.N, NS, 4 * rp.ppw, npark, rp.vec * elem_bytes(p.dtype));
    ResidentChain chain(stream);  // cluster grids of different streams never overlap
    // (the exchange area is taken inside the chain: a context's wrap-around clear is ordered like a launch)
    const ExchangeArea ea = resident_exchange_area(p, kCtlBytes + (size_t)p.N * p.C * NS * 8 + 256, workspace, stream);
    ra.epoch = ea.epoch;
    ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
    const size_t fill_bytes = kCtlBytes + (size_t)p.N * p.C * (NS / 2) * 8;
    // workspace form (large tensors): through a granule region of the context that the previous launch left clean, if there
    // is one — no fill launch (resident_pong_acquire)
    PongArea pong{nullptr, nullptr, 0u, false};
    const bool use_pong = !ea.epoch && resident_pong_acquire(p, fill_bytes, stream, &pong);
    void* area = use_pong ? pong.base : ea.base;
    unsigned* ctl = (unsigned*)area;
    unsigned long long* gran = (unsigned long long*)((char*)area + kCtlBytes);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_pipe_bwd(p.dtype, rp.vec, rp.nv, [&](auto tt, auto vt, auto nt, auto pt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value, PPW = decltype(pt)::value;
        auto launch = [&](auto kern) {
            if (!allow_dynamic_lds(kern, lds)) return;
            const int grid = reshost::grid_for(kern, lds, rp.K, ra.items);
            if (grid < rp.K) return;
            hipError_t e = hipSuccess;
            if (!ea.epoch && (!use_pong || pong.need_fill)) e = hipMemsetAsync(area, 0xff, fill_bytes, stream);
            if (e != hipSuccess) {
                status = (int)e;
                return;
            }
            kern<<<grid, kBlock, lds, stream>>>(ra, npark, (const T*)gy, (const T*)x, (T*)dx, perm, g, f, dg, df, gran, saved,
                                                ctl, pong.clear, pong.clear_qwords, *pin);
            e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
            if (use_pong && e == hipSuccess) resident_pong_commit(p, fill_bytes);
        };
        if (boxed)
            launch(resident_bwd_pipe_kernel<T, VEC, NV, PPW, true>);
        else
            launch(resident_bwd_pipe_kernel<T, VEC, NV, PPW, false>);
    });
    return status;
}

}  // namespace cnsn
