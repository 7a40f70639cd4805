// Host side of the packed small-plane kernels (cnsn_packed_kernels.h): eligibility, run geometry, launches.
#include "cnsn_packed.h"
#include "cnsn_env.h"

#include "cnsn_packed_kernels.h"

namespace cnsn {

namespace {

int gcd_i(int a, int b) {
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    return a;
}

int cu_count_p() {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached = n;
    }
    return cached;
}

template <typename T, bool BOXED, typename Op>
void launch(const PackedGeom& g, const void* in0, const void* in1, const void* in2, void* out0, void* out1,
            const Op& op, hipStream_t stream) {
    const size_t lds = packed_lds_bytes(g, Op::NIN, Op::NSC);
    const int wg_runs = (g.runs + kPackedWaves - 1) / kPackedWaves;
    // enough workgroups to fill the chip several times over, few enough that each amortises its start-up
    int grid = wg_runs;
    const int cap = cu_count_p() * 16;
    if (grid > cap) grid = cap;
    packed_kernel<T, BOXED, Op><<<grid, kBlock, lds, stream>>>(g, (const T*)in0, (const T*)in1, (const T*)in2, (T*)out0,
                                                             (T*)out1, op);
}

// f(TypeTag<T>, bool_constant<BOXED>, IntTag<ADD>)
template <typename F>
void dispatch_p(int dtype, bool boxed, int add, F&& f) {
    auto by_add = [&](auto tt, auto bt) {
        if (add == ADD_PRE)
            f(tt, bt, IntTag<ADD_PRE>{});
        else if (add == ADD_POST)
            f(tt, bt, IntTag<ADD_POST>{});
        else
            f(tt, bt, IntTag<ADD_NONE>{});
    };
    auto by_box = [&](auto tt) {
        if (boxed)
            by_add(tt, IntTag<1>{});
        else
            by_add(tt, IntTag<0>{});
    };
    if (dtype == CNSN_F32)
        by_box(TypeTag<float>{});
    else if (dtype == CNSN_BF16)
        by_box(TypeTag<bf16_t>{});
    else
        by_box(TypeTag<_Float16>{});
}

}  // namespace

bool packed_plan(const Plan& pl, PackedGeom& g) {
    const cnsn_problem_t& p = pl.pr;
    const int b = elem_bytes(p.dtype);
    const int M = p.H * p.W;
    const long long plane_bytes = (long long)M * b;
    if (plane_bytes > 1024) return false;
    // Measured on MI355X (profiles/r01_small_planes.md): the packed kernels win wherever the streaming kernels
    // cannot use full 16-byte vectors (7x7 in any type: 1.4-1.6x; 14x14 bf16: 1.15-1.3x) and lose a little where
    // they can (8x8 / 14x14 / 16x16 fp32), so they take exactly the former.
    if (pl.shape.vec * b >= 16) return false;
    if (const char* e = knob(K_NO_PACKED))
        if (e[0] == '1') return false;
    // planes per run: a multiple of 4 (four 16-lane groups per wave) whose bytes are a multiple of 16,
    // at least ~2 KiB per run and at most 64 planes (one lane fetches the scalars of one plane)
    const int G = 16 / gcd_i(16, (int)(plane_bytes % 16));  // fewest planes whose bytes are a multiple of 16
    int unit = G * 4 / gcd_i(G, 4);  // lcm(G, 4)
    int R = unit;
    while ((long long)R * plane_bytes < 2048 && R + unit <= 64) R += unit;
    if ((long long)R * plane_bytes > 8192) return false;
    g.P = (int)pl.P;
    g.M = M;
    g.Wd = p.W;
    g.R = R;
    g.runs = (int)((pl.P + R - 1) / R);
    g.run_vecs = (int)((long long)R * plane_bytes / 16);
    g.total = (long long)pl.P * plane_bytes;
    g.cb = pl.cb;
    g.sb = pl.sb;
    // the largest instantiation stages 3 tensors + 16 scalars per plane
    if (packed_lds_bytes(g, 3, BC_ROWS + 5) > 64 * 1024) return false;
    return true;
}

void packed_stats(const Plan& pl, const PackedGeom& g, int add, const void* x, const void* addend, double* mom,
                  hipStream_t stream) {
    dispatch_p(pl.pr.dtype, pl.boxed, add == ADD_PRE ? ADD_PRE : ADD_NONE, [&](auto tt, auto bt, auto at) {
        using T = typename decltype(tt)::type;
        constexpr bool BOXED = decltype(bt)::value != 0;
        constexpr int ADD = decltype(at)::value;
        if constexpr (ADD != ADD_POST) {
            PackedStatsOp<T, BOXED, ADD> op{mom, g.P, g.M, pl.mid.Mc, pl.mid.Ms};
            launch<T, BOXED>(g, x, addend, nullptr, nullptr, nullptr, op, stream);
        }
    });
}

void packed_apply_fwd(const Plan& pl, const PackedGeom& g, int add, int relu, const void* x, const void* addend, void* y,
                      const float* coef, hipStream_t stream) {
    dispatch_p(pl.pr.dtype, pl.boxed, add, [&](auto tt, auto bt, auto at) {
        using T = typename decltype(tt)::type;
        constexpr bool BOXED = decltype(bt)::value != 0;
        constexpr int ADD = decltype(at)::value;
        PackedApplyFwdOp<T, BOXED, ADD> op{coef, g.P, relu};
        launch<T, BOXED>(g, x, addend, nullptr, y, nullptr, op, stream);
    });
}

void packed_reduce(const Plan& pl, const PackedGeom& g, int add, int relu, const void* gy, const void* x,
                   const void* addend, const double* saved, float* sums, hipStream_t stream) {
    dispatch_p(pl.pr.dtype, pl.boxed, add, [&](auto tt, auto bt, auto at) {
        using T = typename decltype(tt)::type;
        constexpr bool BOXED = decltype(bt)::value != 0;
        constexpr int ADD = decltype(at)::value;
        PackedReduceOp<T, BOXED, ADD> op{saved, sums, g.P, relu, pl.pr.N, pl.pr.C};
        launch<T, BOXED>(g, gy, x, addend, nullptr, nullptr, op, stream);
    });
}

void packed_apply_bwd(const Plan& pl, const PackedGeom& g, int add, int relu, const void* gy, const void* x,
                      const void* addend, void* dx, void* d_addend, const float* coef, const double* saved,
                      hipStream_t stream) {
    dispatch_p(pl.pr.dtype, pl.boxed, add, [&](auto tt, auto bt, auto at) {
        using T = typename decltype(tt)::type;
        constexpr bool BOXED = decltype(bt)::value != 0;
        constexpr int ADD = decltype(at)::value;
        PackedApplyBwdOp<T, BOXED, ADD> op{coef, saved, g.P, relu, pl.pr.N, pl.pr.C};
        launch<T, BOXED>(g, gy, x, addend, dx, d_addend, op, stream);
    });
}

}  // namespace cnsn
