// Cluster-resident strategy for LARGE planes (1025..4096 vectors of 16 bytes: 128x128 fp32, 128x128 .. 181x181 in 16 bits):
// one plane per workgroup, a quarter per wave (template flag SPLIT of the kernels in cnsn_resident_kernels.h), K = N
// workgroups per channel.  Round 1 ran these planes through the streaming two-pass kernels (segmentation layer 1,
// segmentation/model/cnsn_resnet.py:273-311).  Offered: the op alone (boxed or not, training or inference) and the POST
// add / ReLU epilogue of an un-boxed call (SelfNorm at 'residual'); a PRE add runs two-pass.
#include "cnsn_fused_stream_kernels.h"
#include <optional>
#include "cnsn_resident_host.h"

namespace cnsn {

namespace {

// slots per wave: 8 or 16
inline int split_bucket(int nvec) {
    const int need = (nvec + 255) / 256;
    return need <= 8 ? 8 : (need <= 16 ? 16 : 0);
}

// f(TypeTag<T>, IntTag<VEC>, IntTag<NV>)
template <typename F>
bool dispatch_split(int dtype, int nv, F&& f) {
    auto by_nv = [&](auto tt, auto vt) -> bool {
        if (nv == 8) {
            f(tt, vt, IntTag<8>{});
            return true;
        }
        if (nv == 16) {
            f(tt, vt, IntTag<16>{});
            return true;
        }
        return false;
    };
    if (dtype == CNSN_F32) return by_nv(TypeTag<float>{}, IntTag<4>{});
    if (dtype == CNSN_BF16) return by_nv(TypeTag<bf16_t>{}, IntTag<8>{});
    if (dtype == CNSN_F16) return by_nv(TypeTag<_Float16>{}, IntTag<8>{});
    return false;
}

}  // namespace

ResPlan resident_split_plan(const cnsn_problem_t& p, bool boxed, bool has_chan_perm, int add, int relu, bool backward) {
    ResPlan rp{false, 0, 0, 0, 0};
    (void)relu;
    if (p.strategy == CNSN_STRATEGY_TWO_PASS || p.strategy == CNSN_STRATEGY_LOCAL || p.strategy == CNSN_STRATEGY_MONO ||
        has_chan_perm)
        return rp;
    if (resident_degraded()) return rp;
    if (add == ADD_PRE || (add == ADD_POST && boxed)) return rp;
    const int M = p.H * p.W;
    rp.vec = 16 / elem_bytes(p.dtype);
    if ((boxed ? p.W : M) % rp.vec) return rp;  // 16-byte vectors (inside one row when the call has boxes)
    const int nvec = M / rp.vec;
    if (nvec <= 1024) return rp;  // the one-plane-per-wave kernels take those
    rp.nv = split_bucket(nvec);
    if (rp.nv == 0) return rp;
    rp.ppw = 1;
    rp.K = p.N;  // one plane per workgroup
    if (res_lds_bytes(p.N, 6, 1, BC_ROWS, true) > 64 * 1024) return rp;
    if (rp.K > 2 * reshost::cu_count() || rp.K < 2) return rp;
    if (p.strategy == CNSN_STRATEGY_AUTO) {
        if (!resident_auto_enabled()) return rp;
        const int need = (nvec + 255) / 256;
        if ((rp.nv - need) * 4 > need) return rp;  // a register bucket more than 25 % too large is not worth it
        if (p.dtype != CNSN_F32 && rp.nv > 8) return rp;  // 16-bit, 16 slots: spills (not measured)
        // 16-bit with crop boxes: two-pass won while the region select branched per element (profiles/r02_auto_audit.md); since
        // it is branch-free (profiles/r04_boxed_sweep.md, 128x128 bf16 at (16,256) and (64,64)): the backward -4..-6 %, CrossNorm
        // alone forward -15 %; the forward with SelfNorm stays two-pass (0.094 vs 0.096-0.098 ms)
        if (p.dtype != CNSN_F32 && boxed && !backward && p.sn_active) return rp;
    }
    rp.ok = true;
    return rp;
}

int resident_split_forward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                           const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                           double* saved, void* workspace, hipStream_t stream) {
    const ResPlan rp = resident_split_plan(p, boxed, false, add, relu, false);
    if (!rp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    ResArgs ra = reshost::make_args(p, cb, sb, mid, rp);
    const bool solo = !boxed && !p.cn_active && !(p.sn_active && p.sn_training);
    const bool post = add == ADD_POST, epi = post || relu;
    const int NG = boxed ? 6 : 2;
    const size_t lds = res_lds_bytes(p.N, NG, 1, FC_ROWS, false);
    std::optional<ResidentChain> chain;  // cluster grids of different streams never overlap
    if (!solo) chain.emplace(stream);  // (the exchange area is taken inside the chain: a context's wrap-around clear is ordered like a launch)
    const ExchangeArea ea = solo ? ExchangeArea{workspace, 0u}
                                 : resident_exchange_area(p, kCtlBytes + (size_t)p.N * p.C * NG * 8, workspace, stream);
    ra.epoch = ea.epoch;
    ra.ctl_idle = ea.epoch ? 0u : kCtlIdle;
    const size_t fill_bytes = kCtlBytes + (size_t)p.N * p.C * (NG / 2) * 8;
    PongArea pong{nullptr, nullptr, 0u, false};  // (a granule region of the context instead of workspace + fill: resident_pong_acquire)
    const bool use_pong = !solo && !ea.epoch && resident_pong_acquire(p, fill_bytes, stream, &pong);
    void* area = use_pong ? pong.base : ea.base;
    unsigned* ctl = (unsigned*)area;
    unsigned long long* gran = (unsigned long long*)((char*)area + kCtlBytes);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_split(p.dtype, rp.nv, [&](auto tt, auto vt, auto nt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value;
        auto launch = [&](auto kern) {
            const int grid = reshost::grid_for(kern, lds, rp.K, ra.items);
            if (grid < rp.K) return;
            hipError_t e = hipSuccess;
            if (solo) {
                kern<<<grid, kBlock, lds, stream>>>(ra, (const T*)x, (T*)y, perm, g, f, gran, saved, ctl, (const T*)addend, relu,
                                                    nullptr, 0u, *pin);
            } else {
                if (!ea.epoch && (!use_pong || pong.need_fill)) e = hipMemsetAsync(area, 0xff, fill_bytes, stream);
                if (e != hipSuccess) {
                    status = (int)e;
                    return;
                }
                kern<<<grid, kBlock, lds, stream>>>(ra, (const T*)x, (T*)y, perm, g, f, gran, saved, ctl, (const T*)addend, relu,
                                                    pong.clear, pong.clear_qwords, *pin);
                if (use_pong && hipPeekAtLastError() == hipSuccess) resident_pong_commit(p, fill_bytes);
            }
            e = hipGetLastError();
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        //                         T  VEC NV PPW BOXED  EPI    SOLO   POST   SPLIT
        if (post) {
            if (solo)
                launch(resident_fwd_kernel<T, VEC, NV, 1, false, true, true, true, true>);
            else
                launch(resident_fwd_kernel<T, VEC, NV, 1, false, true, false, true, true>);
        } else if (epi) {  // ReLU alone
            if (boxed)
                launch(resident_fwd_kernel<T, VEC, NV, 1, true, true, false, false, true>);
            else if (solo)
                launch(resident_fwd_kernel<T, VEC, NV, 1, false, true, true, false, true>);
            else
                launch(resident_fwd_kernel<T, VEC, NV, 1, false, true, false, false, true>);
        } else {
            if (boxed)
                launch(resident_fwd_kernel<T, VEC, NV, 1, true, false, false, false, true>);
            else if (solo)
                launch(resident_fwd_kernel<T, VEC, NV, 1, false, false, true, false, true>);
            else
                launch(resident_fwd_kernel<T, VEC, NV, 1, false, false, false, false, true>);
        }
    });
    return status;
}

int resident_split_backward(const cnsn_problem_t& p, Box cb, Box sb, bool boxed, const MidArgs& mid, int add, int relu,
                            const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f,
                            const double* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                            hipStream_t stream) {
    const ResPlan rp = resident_split_plan(p, boxed, false, add, relu, true);
    if (!rp.ok) return CNSN_E_UNSUPPORTED;
    PermInline* pin = perm_inline_scratch();
    if (const int ps = perm_inline_fill(p, perm, pin)) return ps;
    const bool post = add == ADD_POST && relu;  // (POST without ReLU: the plain backward, grad of the addend = grad_y)
    if (post && !d_addend) return CNSN_E_UNSUPPORTED;
    ResArgs ra = reshost::make_args(p, cb, sb, mid, rp);
    const int NS = boxed ? 4 : 2;
    const size_t lds = res_lds_bytes(p.N, NS, 1, BC_ROWS, true);
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
    dispatch_split(p.dtype, rp.nv, [&](auto tt, auto vt, auto nt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value, NV = decltype(nt)::value;
        auto launch = [&](auto kern) {
            const int grid = reshost::grid_for(kern, lds, rp.K, ra.items);
            if (grid < rp.K) return;
            hipError_t e = hipSuccess;
            if (!ea.epoch && (!use_pong || pong.need_fill)) e = hipMemsetAsync(area, 0xff, fill_bytes, stream);
            if (e != hipSuccess) {
                status = (int)e;
                return;
            }
            kern<<<grid, kBlock, lds, stream>>>(ra, (const T*)gy, (const T*)x, (T*)dx, perm, g, f, dg, df, gran, saved, ctl,
                                                (const T*)(post ? addend : nullptr), relu, (T*)d_addend, pong.clear, pong.clear_qwords, *pin);
            e = hipGetLastError();
            if (use_pong && e == hipSuccess) resident_pong_commit(p, fill_bytes);
            status = e == hipSuccess ? CNSN_OK : (int)e;
        };
        //                         T  VEC NV PPW BOXED  EPI    POST   SPLIT
        if (post)
            launch(resident_bwd_kernel<T, VEC, NV, 1, false, true, true, true>);
        else if (relu) {
            if (boxed)
                launch(resident_bwd_kernel<T, VEC, NV, 1, true, true, false, true>);
            else
                launch(resident_bwd_kernel<T, VEC, NV, 1, false, true, false, true>);
        } else {
            if (boxed)
                launch(resident_bwd_kernel<T, VEC, NV, 1, true, false, false, true>);
            else
                launch(resident_bwd_kernel<T, VEC, NV, 1, false, false, false, true>);
        }
    });
    return status;
}

}  // namespace cnsn
