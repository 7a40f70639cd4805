// Channels-last ("NHWC") two-pass path of the op: cnsn_nhwc_kernels.h around the mid kernels of the two-pass strategy.
#include "cnsn_nhwc.h"

#include "cnsn_nhwc_kernels.h"

namespace cnsn {

namespace {

constexpr int kTargetBlocks = 2048;  // workgroups a launch aims for (256 CUs x 8)

int vec_of(int dtype) { return 16 / elem_bytes(dtype); }

NhwcGeom make_nhwc_geom(const Plan& pl) {
    const cnsn_problem_t& p = pl.pr;
    NhwcGeom g;
    g.N = p.N;
    g.C = p.C;
    g.M = p.H * p.W;
    g.tc = p.C / vec_of(p.dtype);
    g.tcb = g.tc < kBlock ? g.tc : kBlock;
    g.rows = kBlock / g.tcb;
    g.ncb = (g.tc + g.tcb - 1) / g.tcb;
    // pixel chunks: enough workgroups to fill the chip, at least 8 pixels per thread and chunk where the plane allows it
    const long per_instance = g.ncb;
    int S = (int)((kTargetBlocks + (long)g.N * per_instance - 1) / ((long)g.N * per_instance));
    const int s_max = g.M / (8 * g.rows) > 0 ? g.M / (8 * g.rows) : 1;
    if (S > s_max) S = s_max;
    if (S < 1) S = 1;
    g.mchunk = (g.M + S - 1) / S;
    g.S = (g.M + g.mchunk - 1) / g.mchunk;
    g.P = pl.P;
    return g;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

template <typename F>
bool dispatch_nhwc(int dtype, F&& f) {
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
void with_add3(int add, F&& f) {
    if (add == ADD_PRE)
        f(IntTag<ADD_PRE>{});
    else if (add == ADD_POST)
        f(IntTag<ADD_POST>{});
    else
        f(IntTag<ADD_NONE>{});
}

}  // namespace

bool nhwc_supported(const Plan& pl, bool has_chan_perm) {
    const cnsn_problem_t& p = pl.pr;
    if (pl.boxed || has_chan_perm) return false;
    if (p.C % vec_of(p.dtype) != 0) return false;
    if (p.H * p.W < 2) return false;
    return true;
}

size_t nhwc_extra_bytes(const Plan& pl) {
    const NhwcGeom g = make_nhwc_geom(pl);
    // part | kshift (forward) / rows (backward) | the common `saved` record a SelfNorm-only backward expands the slim one into
    return align256((size_t)g.S * 2 * g.P * 4) + align256(4 * g.P * 4) + (nhwc_slim_record(pl) ? align256(saved_doubles_of(pl) * 8) : 0) +
           256;
}

size_t nhwc_workspace_bytes(const Plan& pl) {
    if (!nhwc_supported(pl, false)) return 0;
    const size_t two_pass = align256(workspace_bytes_of(pl)) + nhwc_extra_bytes(pl);
    size_t fused = nhwc_slim_record(pl) ? nhwc_fused_extra_bytes(pl) : 0;
    if (nhwc_slim_record(pl) && pl.pr.N <= kBlock && pl.pr.sn_training) {  // (+ the BatchNorm2d-in-front launches: five rows of partial sums)
        const size_t bn = nhwc_bnhead_extra_bytes(pl);
        fused = fused > bn ? fused : bn;
    }
    return two_pass > fused ? two_pass : fused;
}

int nhwc_forward(Plan& pl, int add, int relu, const void* x, const void* addend, const int64_t* perm, GateDev g, GateDev f, void* y,
                 float* saved, void* workspace, size_t workspace_bytes, hipStream_t stream, void* sum_out) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_supported(pl, false)) return CNSN_E_UNSUPPORTED;
    if (p.cn_active && !perm) return CNSN_E_UNSUPPORTED;  // (the mid kernels read the device array)
    if (nhwc_fused_ok(pl)) {  // SelfNorm (+ epilogue) in ONE launch: cnsn_nhwc_fused_kernels.h
        const int st = nhwc_fused_forward(pl, add, relu, x, addend, g, y, saved, workspace, workspace_bytes, stream, sum_out);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    const size_t base = align256(workspace_bytes_of(pl));
    if (workspace_bytes < base + nhwc_extra_bytes(pl)) return CNSN_E_WORKSPACE;
    const NhwcGeom ng = make_nhwc_geom(pl);
    const bool slim = nhwc_slim_record(pl);  // `saved` holds the slim record: the mid kernel's own goes to the workspace
    pl.mid.save_coefs = (relu && saved && !slim) ? 1 : 0;
    const size_t P = pl.P;
    double* mom = (double*)workspace;
    double* saved_d = (saved && !slim) ? (double*)saved : mom + 6 * P;
    float* coef = (float*)(mom + 6 * P + saved_doubles_of(pl));
    float* part = (float*)((char*)workspace + base);
    float* kshift = (float*)((char*)part + align256((size_t)ng.S * 2 * P * 4));
    const int blocks = ng.N * ng.S * ng.ncb;
    const int pblocks = (int)((P + kBlock - 1) / kBlock);
    dispatch_nhwc(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        const size_t lds = (size_t)2 * ng.rows * ng.tcb * VEC * 4;
        if (add == ADD_PRE)  // (keeps X = x + addend in sum_out when the caller asked for it: the apply pass then reads ONE tensor)
            nhwc_stats_kernel<T, VEC, ADD_PRE><<<blocks, kBlock, lds, stream>>>((const T*)x, (const T*)addend, ng, part, kshift,
                                                                                 (T*)sum_out);
        else
            nhwc_stats_kernel<T, VEC, ADD_NONE><<<blocks, kBlock, lds, stream>>>((const T*)x, nullptr, ng, part, kshift, nullptr);
    });
    // (merging the pixel chunks inside the mid kernel instead of by this launch was measured: the finishing kernel takes 5-7 us on
    // all compute units, the same loads inside mid_fwd_kernel's 128-256 workgroups cost it 5-13 us: profiles/r05_mid_blocks.md)
    nhwc_finish_stats_kernel<<<pblocks, kBlock, 0, stream>>>(part, kshift, ng.S, P, ng.M, mom);
    launch_mid_fwd(pl, mom, perm, nullptr, g, f, coef, saved_d, stream);
    if (slim && saved) nhwc_slim_from_saved(pl, saved_d, saved, stream);
    ApplyCoef cf{coef + FC_A_IN * P, coef + FC_XR * P, coef + FC_B_IN * P, coef + FC_A_OUT * P, coef + FC_B_OUT * P};
    dispatch_nhwc(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        if (add == ADD_PRE && sum_out) {
            nhwc_apply_fwd_kernel<T, VEC, ADD_NONE><<<blocks, kBlock, 0, stream>>>((const T*)sum_out, nullptr, (T*)y, ng, cf, relu);
            return;
        }
        with_add3(add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            nhwc_apply_fwd_kernel<T, VEC, ADD><<<blocks, kBlock, 0, stream>>>((const T*)x, (const T*)addend, (T*)y, ng, cf, relu);
        });
    });
    return launch_status();
}

int nhwc_backward(Plan& pl, int add, int relu, const void* gy, const void* x, const void* addend, const int64_t* perm, GateDev g,
                  GateDev f, const float* saved, void* dx, void* d_addend, GateGradDev dg, GateGradDev df, void* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
    const cnsn_problem_t& p = pl.pr;
    if (!nhwc_supported(pl, false)) return CNSN_E_UNSUPPORTED;
    if (p.cn_active && !perm) return CNSN_E_UNSUPPORTED;
    if (add == ADD_POST && relu && !d_addend) return CNSN_E_NULL;
    if (nhwc_fused_ok(pl)) {
        const int st = nhwc_fused_backward(pl, add, relu, gy, x, addend, g, saved, dx, d_addend, dg, workspace, workspace_bytes, stream);
        if (st != CNSN_E_UNSUPPORTED) return st;
    }
    const size_t base = align256(workspace_bytes_of(pl));
    if (workspace_bytes < base + nhwc_extra_bytes(pl)) return CNSN_E_WORKSPACE;
    const NhwcGeom ng = make_nhwc_geom(pl);
    const size_t P = pl.P;
    double* tmp = (double*)workspace;
    float* sums = (float*)(tmp + BT_ROWS * P);
    float* coef = sums + 4 * P;
    const bool slim = nhwc_slim_record(pl);
    float* part = (float*)((char*)workspace + base);
    float* rows = (float*)((char*)part + align256((size_t)ng.S * 2 * P * 4));
    double* expanded = (double*)((char*)rows + align256(4 * P * 4));  // (slim only)
    const double* saved_d = slim ? expanded : (const double*)saved;
    const int blocks = ng.N * ng.S * ng.ncb;
    const int pblocks = (int)((P + kBlock - 1) / kBlock);
    // the backward of an epilogue without ReLU and without PRE add is the plain backward
    const int eff_add = (relu || add == ADD_PRE) ? add : ADD_NONE;
    if (slim)
        nhwc_saved_from_slim(pl, saved, relu, expanded, rows, stream);
    else
        nhwc_saved_rows_kernel<<<pblocks, kBlock, 0, stream>>>(saved_d, p.N, p.C, relu, rows);
    dispatch_nhwc(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        const size_t lds = (size_t)2 * ng.rows * ng.tcb * VEC * 4;
        with_add3(eff_add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            nhwc_bwd_reduce_kernel<T, VEC, ADD><<<blocks, kBlock, lds, stream>>>((const T*)gy, (const T*)x, (const T*)addend, ng, rows,
                                                                                 relu, part);
        });
    });
    nhwc_finish_sums_kernel<<<pblocks, kBlock, 0, stream>>>(part, ng.S, P, sums);
    launch_mid_bwd(pl, sums, saved_d, perm, nullptr, g, f, dg, df, tmp, coef, stream);
    dispatch_nhwc(p.dtype, [&](auto tt, auto vt) {
        using T = typename decltype(tt)::type;
        constexpr int VEC = decltype(vt)::value;
        with_add3(eff_add, [&](auto at) {
            constexpr int ADD = decltype(at)::value;
            nhwc_apply_bwd_kernel<T, VEC, ADD><<<blocks, kBlock, 0, stream>>>((const T*)gy, (const T*)x, (const T*)addend, (T*)dx,
                                                                                (T*)d_addend, ng, coef, rows, relu);
        });
    });
    return launch_status();
}

}  // namespace cnsn
