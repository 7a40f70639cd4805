// Channels-last single-launch kernels (cnsn_nhwc_fused_kernels.h): host side.
#include "cnsn_nhwc.h"

#include "cnsn_nhwc_bnhead_kernels.h"
#include "cnsn_nhwc_fused_kernels.h"
#include "cnsn_resident_host.h"

namespace cnsn {

namespace {

int vec_of(int dtype) { return 16 / elem_bytes(dtype); }
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// 0 never, 1 the AUTO rule (default), 2 wherever the kernels apply, n > 2: AUTO for tensors of at most n MiB (CNSN_NHWC_FUSED)
int fused_mode() {
    const char* e = knob(K_NHWC_FUSED);
    if (!e) return 1;
    const int v = atoi(e);
    return v < 0 ? 0 : v;
}

template <typename F>
bool dispatch_t(int dtype, F&& f) {
    if (dtype == CNSN_F32) {
        f(TypeTag<float>{}, IntTag<4>{});
        return true;
    }
    if (dtype == CNSN_BF16) {
        f(TypeTag<bf16_t>{}, IntTag<8>{});
        return true;
    }
    if (dtype == CNSN_F16) {
        f(TypeTag<_Float16>{}, IntTag<8>{});
        return true;
    }
    return false;
}

template <typename F>
void with_add(int add, F&& f) {
    if (add == ADD_PRE)
        f(IntTag<ADD_PRE>{});
    else if (add == ADD_POST)
        f(IntTag<ADD_POST>{});
    else
        f(IntTag<ADD_NONE>{});
}

NhwcFusedArgs make_args(const Plan& pl, const NhwcGeom& ng, int relu, int gc) {
    const cnsn_problem_t& p = pl.pr;
    NhwcFusedArgs a{};
    a.g = ng;
    a.ntiles = ng.N * ng.S * ng.ncb;
    a.ngroups = p.C / gc;
    a.training = p.sn_training ? 1 : 0;
    a.relu = relu;
    a.keep = 0;
    a.eps_sn = p.eps_sn;
    a.eps_bn = p.eps_bn;
    a.momentum = p.momentum;
    a.inv_n = pl.mid.inv_n;
    a.unbias_n = pl.mid.unbias_n;
    a.bar.host_flag = resident_host_flag();
    a.bar.wait_ticks = resident_wait_ticks();
    const char* fi = knob(K_FAULT_INJECT);
    a.bar.fault = (fi && fi[0] == '1') ? 1 : 0;
    a.bar.ctl_idle = 0u;
    return a;
}

// issue `kern` with a co-resident grid (a multiple of 8: the barrier's groups are equal); the barriers are booked on the
// context before the launch.  `a`: the kernel's first argument (NhwcFusedArgs, or a struct that holds one), `fa` the
// NhwcFusedArgs inside it
template <typename A, typename Kern, typename... Args>
int launch_fused(const Plan& pl, Kern kern, size_t lds, A& a, NhwcFusedArgs& fa, void* ws_bar, hipStream_t stream, Args... args) {
    const int grid = reshost::grid_for(kern, lds, 8, fa.ntiles & ~7);
    if (grid < 8) return CNSN_E_UNSUPPORTED;
    ResidentChain chain(stream);  // persistent grids of different streams never overlap
    const BarArea ba = resident_bar_area(pl.pr, ws_bar, stream, grid, 2);
    if (ba.need_fill) {
        const hipError_t e = hipMemsetAsync(ws_bar, 0, kBarBlock, stream);
        if (e != hipSuccess) return (int)e;
    }
    fa.bar.ctl = ba.ctl;
    fa.bar.block = ba.block;
    fa.bar.group_base = ba.group_base;
    fa.bar.bar_base = ba.bar_base;
    kern<<<grid, kBlock, lds, stream>>>(a, args...);
    return launch_status();
}

}  // namespace

NhwcGeom nhwc_fused_geom(const Plan& pl) {
    const cnsn_problem_t& p = pl.pr;
    NhwcGeom g;
    g.N = p.N;
    g.C = p.C;
    g.M = p.H * p.W;
    g.tc = p.C / vec_of(p.dtype);
    // tiles: two per workgroup of the persistent grid (CNSN_NHWC_WG_PER_CU x the compute units).  First narrower column blocks
    // — down to 64 vector columns: a wave still reads 1 KB of one pixel's row at a time, and nothing is paid for the split —
    // then pixel chunks, each of which costs a row of partial sums per plane (8 floats a chunk and plane against a 7x7 plane's
    // 49 elements): at least 8 pixels per thread and chunk where the plane allows it.
    const long target = 2l * CNSN_NHWC_WG_PER_CU * reshost::cu_count();
    g.tcb = g.tc < kBlock ? g.tc : kBlock;
    while (g.tcb > 64 && g.tcb % 2 == 0 && (long)g.N * ((g.tc + g.tcb - 1) / g.tcb) < target) g.tcb /= 2;
    g.rows = kBlock / g.tcb;
    g.ncb = (g.tc + g.tcb - 1) / g.tcb;
    const long per_chunk = (long)g.N * g.ncb;
    int S = (int)((target + per_chunk - 1) / per_chunk);
    const int s_max = g.M / (8 * g.rows) > 0 ? g.M / (8 * g.rows) : 1;
    if (S > s_max) S = s_max;
    if (S < 1) S = 1;
    g.mchunk = (g.M + S - 1) / S;
    g.S = (g.M + g.mchunk - 1) / g.mchunk;
    g.P = pl.P;
    return g;
}

bool nhwc_slim_record(const Plan& pl) {
    const cnsn_problem_t& p = pl.pr;
    return p.layout == CNSN_LAYOUT_NHWC && !p.cn_active && p.sn_active && !p.sn_two;
}

bool nhwc_fused_ok(const Plan& pl, bool check_health) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_slim_record(pl) || !nhwc_supported(pl, false)) return false;
    if (p.strategy == CNSN_STRATEGY_TWO_PASS || p.strategy == CNSN_STRATEGY_LOCAL || p.strategy == CNSN_STRATEGY_MONO) return false;
    if (check_health && resident_degraded()) return false;  // a persistent launch gave up and nobody re-armed since: not even when forced
    const int mode = fused_mode();
    if (mode == 0) return false;
    if (p.N > kBlock || p.C % CNSN_NHWC_GC != 0 || p.H * p.W < 2) return false;  // (phase B: a thread per instance)
    if (check_health && p.strategy == CNSN_STRATEGY_AUTO && !resident_auto_enabled()) return false;
    if (mode > 2 && p.strategy == CNSN_STRATEGY_AUTO && pl.P * (size_t)(p.H * p.W) * elem_bytes(p.dtype) > ((size_t)mode << 20)) return false;
    return true;
}

size_t nhwc_fused_extra_bytes(const Plan& pl) {
    const NhwcGeom g = nhwc_fused_geom(pl);
    return align256((size_t)g.S * 2 * g.P * 4) + align256(4 * g.P * 4) + kBarBlock + 256;  // part | kshift, gate / cX, c0 | barrier block
}

int nhwc_fused_forward(Plan& pl, int add, int relu, const void* x, const void* addend, GateDev gg, void* y, float* saved,
                       void* workspace, size_t workspace_bytes, hipStream_t stream, void* sum_out) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_fused_ok(pl)) return CNSN_E_UNSUPPORTED;
    if (add != ADD_NONE && !addend) return CNSN_E_NULL;
    if (workspace_bytes < nhwc_fused_extra_bytes(pl)) return CNSN_E_WORKSPACE;
    const NhwcGeom ng = nhwc_fused_geom(pl);
    const size_t P = pl.P;
    NhwcFusedArgs a = make_args(pl, ng, relu, CNSN_NHWC_GC);
    a.part = (float*)workspace;
    a.kshift = (float*)((char*)workspace + align256((size_t)ng.S * 2 * P * 4));
    a.slim = saved;
    a.sum_out = add == ADD_PRE ? sum_out : nullptr;
    a.gout = saved ? saved + (size_t)SL_G * P : a.kshift + P;
    // the first read with the default cache policy when the second one can find it on chip (what phase A reads within reach of
    // the 256 MiB Infinity Cache); non-temporal like every other single-use access beyond that
    a.keep = (size_t)(add == ADD_PRE ? 2 : 1) * P * ng.M * elem_bytes(p.dtype) <= ((size_t)320 << 20) ? 1 : 0;
    void* ws_bar = (char*)a.kshift + align256(4 * P * 4);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_t(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        const size_t lds = (size_t)2 * ng.rows * ng.tcb * VEC * 4;
        if (a.sum_out) {  // (the second read is of what phase A wrote: the first one need not stay in the caches)
            status = launch_fused(pl, nhwc_fused_fwd_kernel<T, VEC, ADD_PRE, false, true>, lds, a, a, ws_bar, stream, (const T*)x,
                                  (const T*)addend, (T*)y, gg);
            return;
        }
        with_add(add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            if (a.keep)
                status = launch_fused(pl, nhwc_fused_fwd_kernel<T, VEC, ADD, true>, lds, a, a, ws_bar, stream, (const T*)x, (const T*)addend,
                                      (T*)y, gg);
            else
                status = launch_fused(pl, nhwc_fused_fwd_kernel<T, VEC, ADD, false>, lds, a, a, ws_bar, stream, (const T*)x,
                                      (const T*)addend, (T*)y, gg);
        });
    });
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] nhwc single-launch fwd: tiles=%d (S=%d rows=%d tcb=%d) groups=%d keep=%d -> status %d\n", a.ntiles,
                ng.S, ng.rows, ng.tcb, a.ngroups, a.keep, status);
    return status;
}

int nhwc_fused_backward(Plan& pl, int add, int relu, const void* gy, const void* x, const void* addend, GateDev gg,
                        const float* saved, void* dx, void* d_addend, GateGradDev dg, void* workspace, size_t workspace_bytes,
                        hipStream_t stream) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_fused_ok(pl)) return CNSN_E_UNSUPPORTED;
    if (!saved) return CNSN_E_NULL;
    if (add == ADD_POST && relu && !d_addend) return CNSN_E_NULL;
    if (workspace_bytes < nhwc_fused_extra_bytes(pl)) return CNSN_E_WORKSPACE;
    // the backward of an epilogue without ReLU and without PRE add is the plain backward
    const int eff_add = (relu || add == ADD_PRE) ? add : ADD_NONE;
    if (eff_add != ADD_NONE && !addend) return CNSN_E_NULL;
    const NhwcGeom ng = nhwc_fused_geom(pl);
    const size_t P = pl.P;
    NhwcFusedArgs a = make_args(pl, ng, relu, CNSN_NHWC_GC_BWD);
    // (G is one more tensor in flight than forward)
    a.keep = (size_t)(eff_add != ADD_NONE ? 3 : 2) * P * ng.M * elem_bytes(p.dtype) <= ((size_t)320 << 20) ? 1 : 0;
    a.part = (float*)workspace;
    a.coefb = (float*)((char*)workspace + align256((size_t)ng.S * 2 * P * 4));
    a.slim = const_cast<float*>(saved);
    void* ws_bar = (char*)a.coefb + align256(4 * P * 4);
    int status = CNSN_E_UNSUPPORTED;
    dispatch_t(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        const size_t lds = (size_t)2 * ng.rows * ng.tcb * VEC * 4;
        with_add(eff_add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            if (a.keep)
                status = launch_fused(pl, nhwc_fused_bwd_kernel<T, VEC, ADD, true>, lds, a, a, ws_bar, stream, (const T*)gy, (const T*)x,
                                      (const T*)addend, (T*)dx, (T*)d_addend, gg, dg);
            else
                status = launch_fused(pl, nhwc_fused_bwd_kernel<T, VEC, ADD, false>, lds, a, a, ws_bar, stream, (const T*)gy, (const T*)x,
                                      (const T*)addend, (T*)dx, (T*)d_addend, gg, dg);
        });
    });
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] nhwc single-launch bwd: tiles=%d (S=%d rows=%d tcb=%d) groups=%d keep=%d -> status %d\n", a.ntiles,
                ng.S, ng.rows, ng.tcb, a.ngroups, a.keep, status);
    return status;
}

void nhwc_slim_from_saved(const Plan& pl, const double* saved_d, float* slim, hipStream_t stream) {
    const int blocks = (int)((pl.P + kBlock - 1) / kBlock);
    nhwc_slim_from_saved_kernel<<<blocks, kBlock, 0, stream>>>(saved_d, pl.pr.N, pl.pr.C, slim);
}

void nhwc_saved_from_slim(const Plan& pl, const float* slim, int relu, double* saved_d, float* rows, hipStream_t stream) {
    const int blocks = (int)((pl.P + kBlock - 1) / kBlock);
    nhwc_saved_from_slim_kernel<<<blocks, kBlock, 0, stream>>>(slim, pl.pr.N, pl.pr.C, relu, saved_d, rows);
}

// ------------------------------------------------------------------------------------------------------------------------
// the block's last BatchNorm2d in front of the op (cnsn_nhwc_bnhead_kernels.h): y = act(SelfNorm(BatchNorm2d(conv_out) + identity))
// ------------------------------------------------------------------------------------------------------------------------
// check_health false: the BACKWARD of a forward that ran these kernels — there is no other kernel that reads its record, so it
// runs them whatever has happened to the cluster strategy in between (a bounded wait at worst: DESIGN.md section 6)
bool nhwc_bnhead_ok(const Plan& pl, bool check_health) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_fused_ok(pl, check_health) || !p.sn_training || p.N < 2 || p.C % kBnGc != 0) return false;
    const NhwcGeom g = nhwc_fused_geom(pl);
    if ((long)g.N * g.S * g.ncb < 8) return false;                // (a grid of at least one workgroup per barrier group)
    return (size_t)g.S * kBnFwd * g.P * 4 < ((size_t)1 << 31);  // (32-bit byte offsets into the partial sums: CohBuf)
}

size_t nhwc_bnhead_extra_bytes(const Plan& pl) {
    const NhwcGeom g = nhwc_fused_geom(pl);
    // part | kshift (conv), kshift (identity), gate / cX, c0 | e0, e1 (x 2 with a BatchNorm2d on the skip path) | barrier block
    return align256((size_t)g.S * kBnFwd * g.P * 4) + align256(4 * g.P * 4) + align256(4 * (size_t)pl.pr.C * 4) + kBarBlock + 256;
}

namespace {
NhwcBnArgs make_bn_args(const Plan& pl, const NhwcGeom& ng, int relu, const cnsn_bn_tail_t& bn, const cnsn_bn_tail_t* bn2, float* bn_stats,
                        void* workspace) {
    const cnsn_problem_t& p = pl.pr;
    NhwcBnArgs a{};
    a.f = make_args(pl, ng, relu, kBnGc);
    a.bn = BnHeadDev{bn.weight, bn.bias, bn.running_mean, bn.running_var, (long long*)bn.num_batches_tracked, bn.eps, bn.momentum};
    if (bn2)
        a.bn2 = BnHeadDev{bn2->weight, bn2->bias, bn2->running_mean, bn2->running_var, (long long*)bn2->num_batches_tracked, bn2->eps,
                          bn2->momentum};
    const double R = (double)p.N * (double)ng.M;
    a.inv_r = 1.0 / R;
    a.unbias_r = R > 1.0 ? R / (R - 1.0) : 1.0;
    a.bn_stats = bn_stats;
    a.f.part = (float*)workspace;
    float* side = (float*)((char*)workspace + align256((size_t)ng.S * kBnFwd * pl.P * 4));
    a.f.kshift = side;
    a.kshift_b = side + pl.P;
    a.f.gout = side + 2 * pl.P;   // forward without `saved`
    a.f.coefb = side + 2 * pl.P;  // backward: cX, c0
    a.chan = (float*)((char*)side + align256(4 * pl.P * 4));
    return a;
}
void* bn_bar_block(const Plan& pl, const NhwcBnArgs& a) { return (char*)a.chan + align256(4 * (size_t)pl.pr.C * 4); }
}  // namespace

int nhwc_bnhead_forward(Plan& pl, int relu, const cnsn_bn_tail_t& bn, const cnsn_bn_tail_t* bn2, const void* conv_out, const void* identity,
                        GateDev gg, void* y, float* saved, float* bn_stats, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_bnhead_ok(pl) || !bn.training || (bn2 && !bn2->training)) return CNSN_E_UNSUPPORTED;
    if (workspace_bytes < nhwc_bnhead_extra_bytes(pl)) return CNSN_E_WORKSPACE;
    const NhwcGeom ng = nhwc_fused_geom(pl);
    NhwcBnArgs a = make_bn_args(pl, ng, relu, bn, bn2, bn_stats, workspace);
    a.f.slim = saved;
    if (saved) a.f.gout = saved + (size_t)SL_G * pl.P;
    a.f.keep = (size_t)2 * pl.P * ng.M * elem_bytes(p.dtype) <= ((size_t)320 << 20) ? 1 : 0;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_t(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        const size_t lds = (size_t)3 * ng.rows * ng.tcb * VEC * 4;
        auto go = [&](auto kern) {
            status = launch_fused(pl, kern, lds, a, a.f, bn_bar_block(pl, a), stream, (const T*)conv_out, (const T*)identity, (T*)y, gg);
        };
        if (bn2)
            a.f.keep ? go(nhwc_bnhead_fwd_kernel<T, VEC, true, true>) : go(nhwc_bnhead_fwd_kernel<T, VEC, false, true>);
        else
            a.f.keep ? go(nhwc_bnhead_fwd_kernel<T, VEC, true, false>) : go(nhwc_bnhead_fwd_kernel<T, VEC, false, false>);
    });
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] nhwc bn-block fwd: tiles=%d (S=%d rows=%d tcb=%d) groups=%d keep=%d -> status %d\n", a.f.ntiles, ng.S,
                ng.rows, ng.tcb, a.f.ngroups, a.f.keep, status);
    return status;
}

int nhwc_bnhead_backward(Plan& pl, int relu, const cnsn_bn_tail_t& bn, const cnsn_bn_tail_t* bn2, const void* gy, const void* conv_out,
                         const void* identity, GateDev gg, const float* saved, const float* bn_stats, void* d_conv, void* d_identity,
                         GateGradDev dg, float* dbn_w, float* dbn_b, float* dbn2_w, float* dbn2_b, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_bnhead_ok(pl, false) || !bn.training || (bn2 && !bn2->training)) return CNSN_E_UNSUPPORTED;
    if (!saved || !bn_stats || !d_conv || !d_identity || !dbn_w || !dbn_b || (bn2 && (!dbn2_w || !dbn2_b))) return CNSN_E_NULL;
    if (workspace_bytes < nhwc_bnhead_extra_bytes(pl)) return CNSN_E_WORKSPACE;
    const NhwcGeom ng = nhwc_fused_geom(pl);
    NhwcBnArgs a = make_bn_args(pl, ng, relu, bn, bn2, const_cast<float*>(bn_stats), workspace);
    a.f.slim = const_cast<float*>(saved);
    a.dbn_w = dbn_w;
    a.dbn_b = dbn_b;
    a.dbn2_w = dbn2_w;
    a.dbn2_b = dbn2_b;
    a.f.keep = (size_t)3 * pl.P * ng.M * elem_bytes(p.dtype) <= ((size_t)320 << 20) ? 1 : 0;
    int status = CNSN_E_UNSUPPORTED;
    dispatch_t(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        const size_t lds = (size_t)3 * ng.rows * ng.tcb * VEC * 4;
        auto go = [&](auto kern) {
            status = launch_fused(pl, kern, lds, a, a.f, bn_bar_block(pl, a), stream, (const T*)gy, (const T*)conv_out, (const T*)identity,
                                  (T*)d_conv, (T*)d_identity, gg, dg);
        };
        if (bn2)
            a.f.keep ? go(nhwc_bnhead_bwd_kernel<T, VEC, true, true>) : go(nhwc_bnhead_bwd_kernel<T, VEC, false, true>);
        else
            a.f.keep ? go(nhwc_bnhead_bwd_kernel<T, VEC, true, false>) : go(nhwc_bnhead_bwd_kernel<T, VEC, false, false>);
    });
    if (knob(K_DEBUG))
        fprintf(stderr, "[cnsn] nhwc bn-block bwd: tiles=%d (S=%d rows=%d tcb=%d) groups=%d keep=%d -> status %d\n", a.f.ntiles, ng.S,
                ng.rows, ng.tcb, a.f.ngroups, a.f.keep, status);
    return status;
}

}  // namespace cnsn
